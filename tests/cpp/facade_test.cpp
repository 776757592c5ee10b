// Exercises the C++ facade (include/gcsa2_hip/gcsa.hpp) the way the reference's own harness
// exercises gcsa::GCSA / gcsa::LCPArray (benchmark/query_gcsa.cpp:87-169): find -> parent ->
// depth -> count -> locate, plus LF, LF_fast, locate(max_positions), sample accessors.
// Prints one line per result; tests/test_facade.py compares them with the oracle.
//
//   facade_test index.bin patterns.txt
#include <gcsa2_hip/gcsa.hpp>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

typedef std::uint64_t u64;

static std::vector<char> read_blob(std::ifstream& in)
{
  u64 bytes = 0;
  in.read(reinterpret_cast<char*>(&bytes), 8);
  std::vector<char> data(bytes + 16, 0);
  in.read(data.data(), bytes);
  return data;
}

int main(int argc, char** argv)
{
  if(argc < 3) { std::cerr << "usage: facade_test index.bin patterns.txt" << std::endl; return 2; }
  std::ifstream in(argv[1], std::ios::binary);
  u64 h[12];
  in.read(reinterpret_cast<char*>(h), sizeof(h));
  gcsa2_host_view view = {};
  view.path_nodes = h[0]; view.edges = h[1]; view.order = h[2]; view.sigma = h[3]; view.fast_chars = h[4];
  view.sample_count = h[5]; view.sample_width = h[6]; view.extra_values_len = h[7]; view.redundant_len = h[8];
  view.lcp_size = h[9]; view.lcp_branching = h[10]; view.lcp_levels = h[11];
  std::vector<std::vector<char>> blobs;
  for(u64 i = 0; i < 2 + view.sigma + 9; i++) { blobs.push_back(read_blob(in)); }
  size_t b = 0;
  view.char2comp = reinterpret_cast<const std::uint8_t*>(blobs[b++].data());
  view.C = reinterpret_cast<const u64*>(blobs[b++].data());
  std::vector<const u64*> bwt;
  for(u64 c = 0; c < view.sigma; c++) { bwt.push_back(reinterpret_cast<const u64*>(blobs[b++].data())); }
  view.bwt = bwt.data();
  view.edge_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.sampled_path_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.stored_samples = reinterpret_cast<const u64*>(blobs[b++].data());
  view.sample_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.extra_filter_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.extra_values_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.redundant_bits = reinterpret_cast<const u64*>(blobs[b++].data());
  view.lcp_offsets = reinterpret_cast<const u64*>(blobs[b++].data());
  view.lcp_data = reinterpret_cast<const std::uint8_t*>(blobs[b++].data());

  // round trip through the container file: save the view, open the index from the file
  std::string container = std::string(argv[1]) + ".g2hv";
  gcsa::check(gcsa2_host_view_save(&view, container.c_str()), "gcsa2_host_view_save()");
  gcsa::GCSA index(container, 0);
  try { gcsa::GCSA broken(std::string(argv[2]), 0); std::cerr << "invalid container accepted" << std::endl; return 1; }
  catch(const std::runtime_error&) {}
  gcsa::LCPArray lcp(index);
  std::cout << "header " << index.size() << " " << index.edgeCount() << " " << index.order() << " "
            << index.sampleCount() << " " << index.sampleBits() << " " << index.sampledPositions() << " "
            << lcp.size() << " " << lcp.values() << " " << lcp.levels() << " " << lcp.branching() << "\n";

  for(gcsa::size_type k = 0; k <= 6; k++) { std::cout << "kmers " << k << " " << gcsa::countKMers(index, k) << "\n"; }

  std::ifstream pin(argv[2]);
  std::string pattern;
  std::vector<std::string> patterns;
  while(std::getline(pin, pattern)) { patterns.push_back(pattern); }

  std::vector<gcsa::range_type> ranges;
  for(const std::string& p : patterns)
  {
    gcsa::range_type r = index.find(p);                       // container overload
    gcsa::range_type r2 = index.find(p.data(), p.length());   // pointer overload
    if(r != r2) { std::cerr << "find overloads disagree" << std::endl; return 1; }
    std::cout << "find " << r.first << " " << r.second << "\n";
    if(!gcsa::Range::empty(r) && r.second < index.size()) { ranges.push_back(r); }
  }
  for(gcsa::range_type r : ranges)
  {
    gcsa::STNode par = lcp.parent(r);
    std::cout << "parent " << par.sp << " " << par.ep << " " << par.left_lcp << " " << par.right_lcp << " " << par.lcp() << "\n";
    std::cout << "depth " << lcp.depth(par.range()) << "\n";
    gcsa::STNode node = lcp.nodeFor(r);
    std::cout << "nodeFor " << node.left_lcp << " " << node.right_lcp << "\n";
    std::cout << "count " << index.count(r) << "\n";
    std::vector<gcsa::node_type> occ;
    index.locate(r, occ);
    std::cout << "locate";
    for(gcsa::node_type v : occ) { std::cout << " " << v; }
    std::cout << "\n";
    index.locate(r, occ, false, false);
    std::cout << "locate_unsorted";
    for(gcsa::node_type v : occ) { std::cout << " " << v; }
    std::cout << "\n";
    index.locate(r, 3, occ);
    std::cout << "locate_max";
    for(gcsa::node_type v : occ) { std::cout << " " << v; }
    std::cout << "\n";
    std::vector<gcsa::range_type> preds(index.alpha.sigma);
    index.LF_fast(r, preds);
    std::cout << "LF_fast";
    for(gcsa::size_type c = 1; c <= index.alpha.fast_chars; c++) { std::cout << " " << preds[c].first << " " << preds[c].second; }
    std::cout << "\n";
    gcsa::range_type lf = index.LF(r, 1);
    std::cout << "LF " << lf.first << " " << lf.second << " " << index.LF(r.first) << "\n";
    std::cout << "sample " << index.sampled(r.first) << " " << index.firstSample(r.first) << " "
              << index.sample(0) << " " << index.lastSample(0) << "\n";
    std::cout << "sv " << lcp.psv(r.first).first << " " << lcp.nsv(r.first).first << " " << lcp.rmq(r).first << " " << lcp[r.first] << "\n";
  }

  // ---- engine additions of round 4: packed k-mers, break points, re-shaping the image ----------------------------------
  // packed find(): the patterns of one length over ACGT, as 2-bit codes, must give what find() gives
  const std::size_t klen = 6;
  std::vector<std::uint8_t> flat;
  std::vector<gcsa::range_type> want;
  for(const std::string& p : patterns)
  {
    if(p.length() < klen || p.find_first_not_of("ACGT") != std::string::npos) { continue; }
    flat.insert(flat.end(), p.begin(), p.begin() + klen);
    want.push_back(index.find(p.substr(0, klen)));
  }
  std::vector<gcsa::range_type> packed = index.find_packed_batch(gcsa::GCSA::pack_kmers(flat.data(), want.size(), klen), klen);
  // (packKMers goes through the index's own alphabet; on the default alphabet it equals the static form.  An empty batch of
  // break points is no records, not an error: ADVICE r04)
  const bool member_same = index.packKMers(flat.data(), want.size(), klen) == gcsa::GCSA::pack_kmers(flat.data(), want.size(), klen);
  std::vector<gcsa::size_type> no_offsets, empty_boff;
  std::vector<gcsa2_break> empty_breaks(3);
  index.match_breaks_batch(std::vector<std::uint8_t>(), no_offsets, 0, empty_boff, empty_breaks);
  const bool empty_ok = empty_breaks.empty() && empty_boff.size() == 1 && empty_boff[0] == 0;
  std::cout << "packed " << want.size() << " " << (packed == want && member_same && empty_ok ? "same" : "DIFFERENT") << "\n";

  // break points (left-maximal matches) of every pattern, all of them and those of at least 3 characters
  std::vector<std::uint8_t> all;
  std::vector<gcsa::size_type> offsets(1, 0);
  for(const std::string& p : patterns) { all.insert(all.end(), p.begin(), p.end()); offsets.push_back(all.size()); }
  for(gcsa::size_type min_length : {gcsa::size_type(0), gcsa::size_type(3)})
  {
    std::vector<gcsa::size_type> boff;
    std::vector<gcsa2_break> breaks;
    index.match_breaks_batch(all, offsets, min_length, boff, breaks);
    std::cout << "breaks " << min_length << " " << breaks.size();
    for(gcsa::size_type q = 0; q + 1 < boff.size(); q++)
    {
      std::cout << " |";
      for(gcsa::size_type j = boff[q]; j < boff[q + 1]; j++) { std::cout << " " << breaks[j].position << ":" << breaks[j].length << ":" << breaks[j].sp << ":" << breaks[j].ep; }
    }
    std::cout << "\n";
  }

  // the memory ladder on a live image: every answer stays what it was
  const gcsa::size_type full = index.deviceBytes();
  bool same = true;
  const int shapes[4][3] = { {-1, -1, 0}, {-1, 2, -1}, {0, 0, -1}, {1, 5, 1} };
  for(const int* shape : shapes)
  {
    index.setTables(shape[0], shape[1], shape[2]);
    for(std::size_t q = 0; q < patterns.size() && q < 30; q++)
    {
      gcsa::range_type r = index.find(patterns[q]);
      std::vector<gcsa::node_type> occ;
      if(!gcsa::Range::empty(r) && r.second < index.size()) { index.locate(r, occ); same = same && occ.size() == index.count(r); }
      same = same && (packed.empty() || index.find_packed_batch(gcsa::GCSA::pack_kmers(flat.data(), 1, klen), klen)[0] == want[0]);
    }
  }
  index.trim();
  index.setPipeline(2, 15, 1);               // a small pipeline with sleeping waits: same answers (the shape is the caller's to choose)
  {
    const std::vector<std::uint8_t> one(patterns[0].begin(), patterns[0].end());
    same = same && index.find(patterns[0]) == index.find_batch(one, std::vector<gcsa::size_type>{0, one.size()})[0];
  }
  std::cout << "ladder " << (same ? "same" : "DIFFERENT") << " " << (index.deviceBytes() <= full ? "ok" : "grew") << " " << index.find(patterns[0]).first << "\n";
  return 0;
}
