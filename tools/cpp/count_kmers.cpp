// count_kmers for the MI355X engine: the reference's k-mer counter (benchmark/count_kmers.cpp) as a
// client of the C++ facade.  Same command line and report lines:
//   count_kmers [-f] [-k N] [-N] [-o X] [-s N] base_name [base_name2]
// One index: countKMers(); two: compareKMers(), with -o X the symmetric difference goes to X.left / X.right.
// (-s is accepted and ignored: the device search is breadth-first and needs no seed k-mers.)
// Extra option: --device N.

#include <gcsa2_hip/gcsa.hpp>

#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool load(gcsa::GCSA& index, const std::string& base_name, int device, const char* who, const char* which)
{
  try { index = gcsa::GCSA(base_name + ".gcsa", "", device); }
  catch(const std::exception& e)
  {
    std::cerr << who << "(): Cannot load the " << which << "index from " << base_name << ".gcsa: " << e.what() << std::endl;
    return false;
  }
  return true;
}

} // namespace

int main(int argc, char** argv)
{
  gcsa::size_type k = 16;                       // DEFAULT_K, count_kmers.cpp:35
  gcsa::KMerSearchParameters parameters;
  int device = 0;
  std::vector<std::string> names;
  for(int i = 1; i < argc; i++)
  {
    std::string arg = argv[i];
    if(arg == "-f") { parameters.force = true; }
    else if(arg == "-N") { parameters.include_Ns = true; }
    else if(arg == "-k" && i + 1 < argc) { k = std::stoul(argv[++i]); }
    else if(arg == "-o" && i + 1 < argc) { parameters.output = argv[++i]; }
    else if(arg == "-s" && i + 1 < argc) { parameters.seed_length = std::stoul(argv[++i]); }
    else if(arg == "--device" && i + 1 < argc) { device = std::atoi(argv[++i]); }
    else if(!arg.empty() && arg[0] == '-') { std::cerr << "count_kmers: Unknown option " << arg << std::endl; return EXIT_FAILURE; }
    else { names.push_back(arg); }
  }
  if(argc < 2)
  {
    std::cerr << "Kmer counter (MI355X engine)" << std::endl;
    std::cerr << "usage: count_kmers [options] base_name [base_name2]" << std::endl;
    std::cerr << "  -f    Force counting kmers longer than the order of the index" << std::endl;
    std::cerr << "  -k N  Set the length of the kmers to N (default 16)" << std::endl;
    std::cerr << "  -N    Include kmers containing Ns" << std::endl;
    std::cerr << "  -o X  Output the symmetric difference to X.left and X.right" << std::endl;
    std::cerr << "  -s N  Accepted for compatibility (seed kmers are not needed)" << std::endl << std::endl;
    return EXIT_SUCCESS;
  }
  if(names.empty()) { std::cerr << "count_kmers: Base name not specified" << std::endl; return EXIT_FAILURE; }
  const bool compare = names.size() > 1;

  std::cout << "Kmer counter (MI355X engine)" << std::endl;
  if(compare) { std::cout << "Left name:   " << names[0] << std::endl << "Right name:  " << names[1] << std::endl; }
  else { std::cout << "Base name:   " << names[0] << std::endl; }
  std::cout << "K:           " << k << std::endl;
  std::cout << "Options:     seed=" << parameters.seed_length;
  if(parameters.force) { std::cout << " force"; }
  if(parameters.include_Ns) { std::cout << " include_Ns"; }
  if(!parameters.output.empty()) { std::cout << " output=" << parameters.output; }
  std::cout << std::endl << std::endl;

  if(!compare)
  {
    gcsa::GCSA index;
    if(!load(index, names[0], device, "countKmers", "")) { return EXIT_FAILURE; }
    std::cout << "GCSA:        " << index.size() << " paths, order " << index.order() << std::endl;
    double start = now();
    gcsa::size_type kmer_count = gcsa::countKMers(index, k, parameters);
    double seconds = now() - start;
    std::cout << "Kmers:       " << kmer_count << std::endl << std::endl;
    std::cout << "Kmers counted in " << seconds << " seconds (" << (kmer_count / seconds) << " / s)" << std::endl << std::endl;
    return EXIT_SUCCESS;
  }

  gcsa::GCSA left, right;
  if(!load(left, names[0], device, "compareKmers", "first ")) { return EXIT_FAILURE; }
  std::cout << "Left:        " << left.size() << " paths, order " << left.order() << std::endl;
  if(!load(right, names[1], device, "compareKmers", "second ")) { return EXIT_FAILURE; }
  std::cout << "Right:       " << right.size() << " paths, order " << right.order() << std::endl;
  double start = now();
  std::array<gcsa::size_type, 3> results = gcsa::compareKMers(left, right, k, parameters);
  double seconds = now() - start;
  std::cout << "Shared:      " << results[0] << " kmers" << std::endl;
  std::cout << "Left:        " << results[1] << " unique kmers" << std::endl;
  std::cout << "Right:       " << results[2] << " unique kmers" << std::endl << std::endl;
  std::cout << "Kmers counted in " << seconds << " seconds (" << ((results[0] + results[1] + results[2]) / seconds) << " / s)" << std::endl << std::endl;
  return EXIT_SUCCESS;
}
