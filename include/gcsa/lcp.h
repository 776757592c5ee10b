// <gcsa/lcp.h> of the MI355X engine: gcsa::STNode and gcsa::LCPArray with the reference's public interface
// (reference include/gcsa/lcp.h:40-194, src/lcp.cpp:276-519), forwarded to the device image behind the C ABI.
// An LCPArray is either loaded on its own (default constructor + load(), as the reference's tools do) or a
// view of the LCP part of a GCSA image that was created with its .lcp file (LCPArray(const GCSA&)).
#ifndef GCSA2_HIP_GCSA_LCP_H
#define GCSA2_HIP_GCSA_LCP_H

#include "gcsa.h"

namespace gcsa
{

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

class LCPArray
{
public:
  typedef gcsa::size_type size_type;
  typedef STNode          node_type;

  LCPArray() : handle(nullptr) {}                                            // lcp.cpp:50-52
  LCPArray(const LCPArray& source) = default;                                // copies share the immutable device image
  LCPArray(LCPArray&& source) noexcept { this->take(source); }
  ~LCPArray() = default;
  void swap(LCPArray& another)
  {
    std::swap(header, another.header); std::swap(handle, another.handle); owner.swap(another.owner); host.swap(another.host);
  }
  LCPArray& operator=(const LCPArray& source) = default;
  LCPArray& operator=(LCPArray&& source) noexcept { if(this != &source) { this->take(source); } return *this; }

  inline static const std::string EXTENSION = ".lcp";                        // lcp.cpp:45

  // The LCP part of an index image created together with its .lcp file (engine-specific).
  explicit LCPArray(const GCSA& index) : handle(index.handle), owner(index.shared())
  {
    header.size = (handle != nullptr ? gcsa2_lcp_size(handle) : 0);
    header.branching = (handle != nullptr ? gcsa2_lcp_branching(handle) : 0);
  }

  // LCPArray::load (lcp.cpp:130-143): a stand-alone image of the array on Device::current(); throws
  // std::runtime_error("LCP::load(): Invalid header: ...") like the reference.
  void load(std::istream& in)
  {
    const std::streampos start = in.tellg();
    std::vector<char> data = readRest(in);
    gcsa2_view_storage* storage = nullptr;
    std::uint64_t consumed = 0;
    if(gcsa2_host_view_parse_lcp(data.data(), data.size(), &consumed, &storage) != GCSA2_OK) { throw std::runtime_error(gcsa2_last_error()); }
    std::shared_ptr<gcsa2_view_storage> keep(storage, gcsa2_host_view_free);
    in.clear();
    if(start != std::streampos(-1)) { in.seekg(start + std::streamoff(consumed)); }
    gcsa2_index* raw = nullptr;
    check(gcsa2_lcp_create(gcsa2_host_view_get(storage), Device::current(), &raw), "LCP::load()");
    owner.reset(raw, gcsa2_index_destroy);
    handle = raw;
    header = LCPHeader();
    header.size = gcsa2_lcp_size(raw); header.branching = gcsa2_lcp_branching(raw);
    host = (GCSA::retainHostView() ? keep : nullptr);
  }

  // LCPArray::serialize (lcp.cpp:116-128); needs the retained host view, see GCSA::serialize.
  size_type serialize(std::ostream& out, sdsl::structure_tree_node* = nullptr, std::string = "") const
  {
    if(!host) { throw std::runtime_error("LCP::serialize(): the host view of this array was not retained (GCSA::retainHostView)"); }
    std::uint64_t written = 0;
    check(gcsa2_host_view_serialize_lcp(gcsa2_host_view_get(host.get()), &LCPArray::write_to, &out, &written), "LCP::serialize()");
    return written;
  }

  size_type size() const { return handle != nullptr ? gcsa2_lcp_size(handle) : 0; }
  size_type values() const { return handle != nullptr ? gcsa2_lcp_values(handle) : 0; }
  size_type levels() const { return handle != nullptr ? gcsa2_lcp_levels(handle) : 0; }
  size_type branching() const { return header.branching; }
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

  LCPHeader header;          // lcp.h:182
  gcsa2_index* handle;       // the device image holding the array (nullptr: empty)

private:
  range_type sv(int op, size_type pos) const
  {
    size_type out[2];
    check(gcsa2_sv_batch(handle, op, &pos, 1, out), "LCPArray::psv/nsv()");
    return range_type(out[0], out[1]);
  }
  void take(LCPArray& source)
  {
    header = source.header; handle = source.handle; owner = std::move(source.owner); host = std::move(source.host);
    source.handle = nullptr; source.header = LCPHeader();
  }
  static void write_to(void* stream, const void* data, std::uint64_t bytes)
  {
    static_cast<std::ostream*>(stream)->write(static_cast<const char*>(data), std::streamsize(bytes));
  }
  std::shared_ptr<gcsa2_index> owner;
  std::shared_ptr<gcsa2_view_storage> host;
};

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_LCP_H
