// <gcsa/support.h> of the MI355X engine: node_type / Node and the Alphabet members callers read
// (reference include/gcsa/support.h:93-155, 441-471).  The construction-time parts of the reference's
// support.h (ConstructionParameters, KMer, PathNode, the Sadakane counter classes, ...) are out of scope: the
// counters live inside the device image.
#ifndef GCSA2_HIP_GCSA_SUPPORT_H
#define GCSA2_HIP_GCSA_SUPPORT_H

#include "utils.h"

namespace gcsa
{

typedef std::uint64_t node_type;                          // support.h:441

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
  static std::string decode(node_type node)               // support.cpp:594-602
  {
    return std::to_string(id(node)) + ":" + (rc(node) ? "-" : "") + std::to_string(offset(node));
  }
};

// The members of gcsa::Alphabet that callers read as `index.alpha.*` (support.h:150-151, e.g.
// src/algorithms.cpp:127,369), as plain vectors filled from the device image's copy.
class Alphabet
{
public:
  typedef gcsa::size_type size_type;
  constexpr static size_type MAX_SIGMA = 256;
  constexpr static size_type SOURCE_COMP = 6, SINK_COMP = 0;      // support.h:100-101
  constexpr static size_type FAST_CHARS = 4;                      // support.h:104

  Alphabet() : char2comp(256, 0), comp2char(), C(1, 0), sigma(0), fast_chars(0) {}
  void swap(Alphabet& a) { std::swap(*this, a); }

  std::vector<std::uint8_t> char2comp, comp2char;
  std::vector<size_type>    C;
  size_type                 sigma, fast_chars;

  // from a handle: char2comp and C are the image's; comp2char is the first (upper-case) byte of every comp, which
  // gives "$ACGTN#" for the reference's default alphabet (support.cpp:69-92)
  void read(const gcsa2_index* handle)
  {
    sigma = gcsa2_sigma(handle); fast_chars = gcsa2_fast_chars(handle);
    char2comp.assign(256, 0); C.assign(sigma + 1, 0);
    gcsa2_alphabet(handle, char2comp.data(), C.data());
    comp2char.assign(sigma, 0);
    for(size_type c = 0; c < sigma; c++)
    {
      int first = -1, upper = -1;
      for(int b = 0; b < 256; b++)
      {
        if(char2comp[b] != c) { continue; }
        if(first < 0) { first = b; }
        if(upper < 0 && !(b >= 'a' && b <= 'z') && b != 0) { upper = b; }
      }
      comp2char[c] = std::uint8_t(upper >= 0 ? upper : (first >= 0 ? first : 0));
    }
    const std::string dflt = "$ACGTN#";
    bool is_default = (sigma == dflt.size());
    for(size_type c = 0; is_default && c < sigma; c++) { is_default = (char2comp[std::uint8_t(dflt[c])] == c); }
    if(is_default) { comp2char.assign(dflt.begin(), dflt.end()); }
  }
};

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_SUPPORT_H
