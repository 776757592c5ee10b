// verifyIndex() the way the reference's build_gcsa -v drives it (src/algorithms.cpp:101-295), written against the
// reference's names only: <gcsa/gcsa.h>, <gcsa/lcp.h>, <gcsa/algorithms.h>, gcsa::KMer / Key / Node.
//
//   verify_client base_name kmers.txt k [drop|alter|nolcp]
//
// kmers.txt: one k-mer per line, "LABEL node_value" (label padded with '$' to k characters, node_value = the node_type of
// its start position).  drop: the last start node of the first multi-occurrence label is left out (count() must fail);
// alter: one start node is replaced (locate() must fail); nolcp: lcp == 0, the parent() checks are skipped.
#define GCSA2_HIP_SDSL_IO
#include <gcsa/gcsa.h>
#include <gcsa/lcp.h>
#include <gcsa/algorithms.h>

#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using namespace gcsa;

int main(int argc, char** argv)
{
  if(argc < 4) { std::cerr << "usage: verify_client base_name kmers k [drop|alter|nolcp]" << std::endl; return 2; }
  const std::string base_name = argv[1], mode = (argc > 4 ? argv[4] : "");
  const size_type k = std::stoul(argv[3]);

  GCSA index;
  if(!sdsl::load_from_file(index, base_name + GCSA::EXTENSION)) { std::cerr << "cannot load the index" << std::endl; return 1; }
  LCPArray lcp;
  if(!sdsl::load_from_file(lcp, base_name + LCPArray::EXTENSION)) { std::cerr << "cannot load the LCP array" << std::endl; return 1; }

  std::vector<KMer> kmers;
  {
    std::ifstream in(argv[2]);
    std::string label;
    node_type from;
    while(in >> label >> from) { kmers.push_back(KMer(Key::encode(index.alpha, label, 0, 0), from, 0)); }
  }
  std::cout << "kmers " << kmers.size() << " " << Key::decode(kmers.front().key, k, index.alpha) << std::endl;

  if(mode == "drop")
  {
    for(size_type i = 0; i + 1 < kmers.size(); i++)
    {
      if(Key::label(kmers[i].key) == Key::label(kmers[i + 1].key) && kmers[i].from != kmers[i + 1].from) { kmers.erase(kmers.begin() + i); break; }
    }
  }
  if(mode == "alter") { kmers[kmers.size() / 2].from = Node::encode(Node::id(kmers[kmers.size() / 2].from) + 100000, 3); }

  const bool ok = verifyIndex(index, (mode == "nolcp" ? 0 : &lcp), kmers, k);
  std::cout << "result " << ok << std::endl;
  return ok ? 0 : 3;
}
