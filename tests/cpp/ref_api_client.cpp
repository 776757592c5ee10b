// A caller written against the REFERENCE's names only -- <gcsa/gcsa.h>, <gcsa/lcp.h>, <gcsa/algorithms.h>,
// gcsa::GCSA / LCPArray / STNode / Range / Node, sdsl::load_from_file, index.header, index.alpha -- the way
// benchmark/query_gcsa.cpp:45-179 and a vg-style MEM loop use them.  Nothing in this file mentions the engine.
// Compiled against include/ of this repository and linked with -lgcsa2_hip it runs on the MI355X;
// tests/test_facade.py compares every line it prints with the oracle.
//
//   ref_api_client base_name patterns.txt        (opens base_name.gcsa and base_name.lcp)
#define GCSA2_HIP_SDSL_IO      // this program spells the file helpers sdsl::... and does not link SDSL
#include <gcsa/gcsa.h>
#include <gcsa/lcp.h>
#include <gcsa/algorithms.h>

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

using namespace gcsa;

int main(int argc, char** argv)
{
  if(argc < 3) { std::cerr << "usage: ref_api_client base_name patterns" << std::endl; return 2; }
  std::string base_name = argv[1];

  GCSA::retainHostView(true);                 // engine knob (see <gcsa/gcsa.h>): serialize() below needs the host arrays
  GCSA index;
  std::string gcsa_name = base_name + GCSA::EXTENSION;
  if(!sdsl::load_from_file(index, gcsa_name)) { std::cerr << "cannot load " << gcsa_name << std::endl; return 1; }
  LCPArray lcp;
  std::string lcp_name = base_name + LCPArray::EXTENSION;
  if(!sdsl::load_from_file(lcp, lcp_name)) { std::cerr << "cannot load " << lcp_name << std::endl; return 1; }

  std::cout << "header " << index.header.check() << " " << index.header.version << " " << index.header.path_nodes << " "
            << index.header.edges << " " << index.header.order << " " << index.header.flags << std::endl;
  std::cout << "index " << index.size() << " " << index.edgeCount() << " " << index.order() << " " << index.sampleCount() << " "
            << index.sampleBits() << " " << index.empty() << std::endl;
  std::cout << "alpha " << index.alpha.sigma << " " << index.alpha.fast_chars << " "
            << std::string(index.alpha.comp2char.begin(), index.alpha.comp2char.end());
  for(size_type c = 0; c <= index.alpha.sigma; c++) { std::cout << " " << index.alpha.C[c]; }
  std::cout << std::endl;
  std::cout << "lcp " << lcp.header.check() << " " << lcp.size() << " " << lcp.values() << " " << lcp.levels() << " " << lcp.branching() << std::endl;

  std::vector<std::string> patterns;
  {
    std::ifstream in(argv[2]);
    std::string line;
    while(std::getline(in, line)) { patterns.push_back(line); }
  }

  for(const std::string& pattern : patterns)
  {
    range_type range = index.find(pattern);                                  // gcsa.h:112-116
    range_type again = index.find(pattern.data(), pattern.length());         // gcsa.h:118-122
    // the low-level loop vg drives: charRange, then one LF per character (gcsa.h:101-107)
    range_type manual(0, index.size() - 1);
    if(!pattern.empty())
    {
      manual = index.charRange(index.alpha.char2comp[(unsigned char)pattern.back()]);
      for(size_type i = pattern.length() - 1; i > 0 && !Range::empty(manual); i--)
      {
        manual = index.LF(manual, index.alpha.char2comp[(unsigned char)pattern[i - 1]]);
      }
    }
    std::cout << "find " << range.first << " " << range.second << " " << (again == range) << " " << (manual == range)
              << " " << Range::length(range) << " " << Range::empty(range) << std::endl;
    if(Range::empty(range) || range.second >= index.size()) { continue; }
    STNode parent = lcp.parent(range);                                       // query_gcsa.cpp:112
    std::cout << "parent " << parent.range().first << " " << parent.range().second << " " << parent.lcp() << " "
              << lcp.depth(parent.range()) << " " << (lcp.parent(lcp.nodeFor(range)) == parent) << std::endl;
    std::vector<node_type> occurrences;
    index.locate(range, occurrences);                                        // query_gcsa.cpp:159
    std::cout << "locate " << index.count(range) << " " << occurrences.size();
    for(node_type node : occurrences) { std::cout << " " << Node::decode(node); }
    std::cout << std::endl;
  }
  std::cout << "kmers " << countKMers(index, 3) << std::endl;

  // value semantics (gcsa.cpp:56-138): copy, move, swap
  GCSA copy(index), other;
  other.swap(copy);
  GCSA moved(std::move(other));
  std::cout << "copies " << copy.size() << " " << moved.size() << " " << (moved.find(patterns[0]) == index.find(patterns[0])) << " "
            << copy.find(patterns[0]).first << " " << copy.find(patterns[0]).second + 1 << std::endl;

  // serialize() writes the reference's byte stream: byte-identical to the files that were loaded
  {
    std::ostringstream gcsa_bytes, lcp_bytes;
    size_type a = index.serialize(gcsa_bytes), b = lcp.serialize(lcp_bytes);
    std::ifstream f1(gcsa_name, std::ios_base::binary), f2(lcp_name, std::ios_base::binary);
    std::stringstream s1, s2;
    s1 << f1.rdbuf(); s2 << f2.rdbuf();
    std::cout << "serialize " << (gcsa_bytes.str() == s1.str()) << " " << (a == s1.str().size()) << " "
              << (lcp_bytes.str() == s2.str()) << " " << (b == s2.str().size()) << std::endl;
  }

  // error behaviour: load() throws on an invalid header (gcsa.cpp:188-193, lcp.cpp:134-139)
  try { GCSA broken; std::istringstream in(std::string(64, 'x')); broken.load(in); std::cout << "invalid accepted" << std::endl; return 1; }
  catch(const std::runtime_error& e) { std::cout << "gcsa error " << (std::string(e.what()).find("GCSA::load(): Invalid header") != std::string::npos) << std::endl; }
  try { LCPArray broken; std::istringstream in(std::string(64, 'x')); broken.load(in); std::cout << "invalid accepted" << std::endl; return 1; }
  catch(const std::runtime_error& e) { std::cout << "lcp error " << (std::string(e.what()).find("LCP::load(): Invalid header") != std::string::npos) << std::endl; }
  // two structures in one stream (load() leaves the rest of a seekable stream unread)
  {
    std::ostringstream both;
    index.serialize(both); lcp.serialize(both);
    std::istringstream in(both.str());
    GCSA first; LCPArray second;
    first.load(in); second.load(in);
    std::cout << "stream " << first.size() << " " << second.size() << " " << (first.find(patterns[0]) == index.find(patterns[0])) << std::endl;
  }
  return 0;
}
