// query_gcsa for the MI355X engine: the reference's query benchmark (benchmark/query_gcsa.cpp) as a client
// of the C++ facade.  Same command line (`query_gcsa base_name [patterns]`), same input conventions
// (base_name.gcsa + base_name.lcp; one pattern per line, empty rows skipped, rows consisting of Ns
// dropped: query_gcsa.cpp:53-85,186-204) and the same report lines; every phase is ONE batched call
// (find -> parent -> depth -> count -> locate, query_gcsa.cpp:87-169), timed on the host around it.
//
//   g++ -std=c++17 -O2 -Iinclude tools/cpp/query_gcsa.cpp -Lgcsa2_amd/lib -lgcsa2_hip -o gcsa2_amd/lib/query_gcsa
//
// Options after the positional arguments: --device N, --g2hv (base_name is a G2HV container).

#include <gcsa2_hip/gcsa.hpp>

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace {

const double MEGABYTE = 1048576.0;

void header(const std::string& name, std::size_t indent = 18)      // utils.cpp:113-119
{
  std::cout << name << ":";
  if(name.length() + 1 < indent) { std::cout << std::string(indent - 1 - name.length(), ' '); }
}

void report_time(const std::string& name, gcsa::size_type queries, double seconds)      // utils.cpp:121-127
{
  header(name);
  std::cout << queries << " queries in " << seconds << " seconds (" << (seconds / queries * 1e6) << " µs/query)" << std::endl;
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

int main(int argc, char** argv)
{
  std::vector<std::string> positional;
  int device = 0; bool container = false;
  for(int i = 1; i < argc; i++)
  {
    std::string arg = argv[i];
    if(arg == "--device" && i + 1 < argc) { device = std::atoi(argv[++i]); }
    else if(arg == "--g2hv") { container = true; }
    else { positional.push_back(arg); }
  }
  if(positional.empty())
  {
    std::cerr << "GCSA2 query benchmark (MI355X engine)" << std::endl;
    std::cerr << "usage: query_gcsa base_name [patterns] [--device N] [--g2hv]" << std::endl << std::endl;
    return EXIT_SUCCESS;
  }
  const std::string base_name = positional[0], pattern_name = (positional.size() > 1 ? positional[1] : "");
  std::cout << "GCSA2 query benchmark (MI355X engine)" << std::endl;
  header("Base name"); std::cout << base_name << std::endl;
  if(!pattern_name.empty()) { header("Pattern file"); std::cout << pattern_name << std::endl; }
  std::cout << std::endl;

  gcsa::GCSA index;
  try
  {
    if(container) { index = gcsa::GCSA(base_name, device); }
    else { index = gcsa::GCSA(base_name + ".gcsa", base_name + ".lcp", device); }
  }
  catch(const std::exception& e)
  {
    std::cerr << "query_gcsa: Cannot load the index from " << base_name << (container ? "" : ".gcsa / .lcp") << ": " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
  gcsa::LCPArray lcp(index);

  if(pattern_name.empty())      // printStatistics(), src/algorithms.cpp:620-645 (sizes: the device image)
  {
    header("Paths"); std::cout << index.size() << std::endl;
    header("Edges"); std::cout << index.edgeCount() << std::endl;
    header("Samples"); std::cout << index.sampleCount() << " (at " << index.sampledPositions() << " positions, " << index.sampleBits() << " bits each)" << std::endl;
    header("Max query"); std::cout << index.order() << std::endl << std::endl;
    header("LCP array"); std::cout << lcp.values() << " values in " << lcp.levels() << " levels, branching " << lcp.branching() << std::endl;
    header("Device image"); std::cout << gcsa2_device_bytes(index.handle) / MEGABYTE << " MB in HBM (device " << gcsa2_device(index.handle) << ")" << std::endl;
    std::cout << std::endl;
    return EXIT_SUCCESS;
  }
  header("Device image"); std::cout << gcsa2_device_bytes(index.handle) / MEGABYTE << " MB" << std::endl;

  // Patterns: one per row, empty rows skipped, rows of Ns dropped; concatenated for the batched calls.
  std::vector<std::uint8_t> text;
  std::vector<gcsa::size_type> offsets(1, 0);
  {
    std::ifstream in(pattern_name.c_str(), std::ios_base::binary);
    if(!in) { std::cerr << "query_gcsa: Cannot open pattern file " << pattern_name << std::endl; return EXIT_FAILURE; }
    std::string row;
    while(std::getline(in, row))
    {
      if(row.empty() || row.find_first_not_of('N') == std::string::npos) { continue; }
      text.insert(text.end(), row.begin(), row.end());
      offsets.push_back(text.size());
    }
  }
  const gcsa::size_type patterns = offsets.size() - 1;
  header("Patterns"); std::cout << patterns << " (total " << text.size() / MEGABYTE << " MB)" << std::endl << std::endl;
  if(patterns == 0) { return EXIT_SUCCESS; }

  std::vector<gcsa::range_type> ranges;
  std::vector<gcsa::size_type> lengths;
  {
    double start = now();
    std::vector<gcsa::range_type> all = index.find_batch(text, offsets);
    double seconds = now() - start;
    gcsa::size_type total = 0;
    for(gcsa::size_type i = 0; i < patterns; i++)
    {
      if(!gcsa::Range::empty(all[i])) { ranges.push_back(all[i]); lengths.push_back(offsets[i + 1] - offsets[i]); }
      total += gcsa::Range::length(all[i]);
    }
    report_time("find()", patterns, seconds);
    header("find()");
    std::cout << "Found " << ranges.size() << " patterns matching " << total << " paths (" << (text.size() / MEGABYTE / seconds) << " MB/s)" << std::endl << std::endl;
  }
  if(ranges.empty()) { return EXIT_SUCCESS; }

  std::vector<gcsa::range_type> parents(ranges.size());
  {
    double start = now();
    std::vector<gcsa::STNode> nodes = lcp.parent_batch(ranges);
    double seconds = now() - start;
    gcsa::size_type total = 0;
    for(std::size_t i = 0; i < nodes.size(); i++) { total += lengths[i] - nodes[i].lcp(); parents[i] = nodes[i].range(); }
    report_time("parent()", ranges.size(), seconds);
    header("parent()"); std::cout << "Average distance " << (total / double(ranges.size())) << " characters" << std::endl << std::endl;
  }
  {
    double start = now();
    std::vector<gcsa::size_type> depths = lcp.depth_batch(parents);
    double seconds = now() - start;
    gcsa::size_type total = 0;
    for(std::size_t i = 0; i < depths.size(); i++) { total += lengths[i] - depths[i]; }
    report_time("depth()", ranges.size(), seconds);
    header("depth()"); std::cout << "Average distance " << (total / double(parents.size())) << " characters" << std::endl << std::endl;
  }

  std::vector<gcsa::size_type> counts;
  {
    double start = now();
    counts = index.count_batch(ranges);
    double seconds = now() - start;
    gcsa::size_type total = 0;
    for(gcsa::size_type c : counts) { total += c; }
    report_time("count()", ranges.size(), seconds);
    header("count()"); std::cout << total << " occurrences" << std::endl << std::endl;
  }
  {
    double start = now();
    std::vector<gcsa::size_type> value_offsets;
    std::vector<gcsa::node_type> values;
    index.locate_batch(ranges, value_offsets, values);
    double seconds = now() - start;
    for(std::size_t i = 0; i < ranges.size(); i++) { counts[i] -= value_offsets[i + 1] - value_offsets[i]; }
    report_time("locate()", ranges.size(), seconds);
    header("locate()"); std::cout << values.size() << " occurrences (" << (seconds / values.size() * 1e6) << " µs/occurrence)" << std::endl << std::endl;
  }
  for(gcsa::size_type c : counts)
  {
    if(c != 0) { std::cout << "Warning: count() and locate() returned inconsistent results" << std::endl << std::endl; break; }
  }
  return EXIT_SUCCESS;
}
