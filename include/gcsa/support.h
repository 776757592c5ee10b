// <gcsa/support.h> of the MI355X engine: node_type / Node and the Alphabet members callers read
// (reference include/gcsa/support.h:93-155, 441-471), plus Key / KMer as far as verifyIndex() reads them
// (support.h:378-497).  The construction-time parts of the reference's support.h (ConstructionParameters, PathNode,
// the Sadakane counter classes, ...) are out of scope: the counters live inside the device image.
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

  // from a handle: char2comp and C are the image's; comp2char is derived from char2comp by the rule GCSA::serialize uses too
  // (gcsa2_derive_comp2char: "$ACGTN#" for the reference's default alphabet, support.cpp:69-92)
  void read(const gcsa2_index* handle)
  {
    sigma = gcsa2_sigma(handle); fast_chars = gcsa2_fast_chars(handle);
    char2comp.assign(256, 0); C.assign(sigma + 1, 0);
    gcsa2_alphabet(handle, char2comp.data(), C.data());
    comp2char.assign(sigma, 0);
    gcsa2_derive_comp2char(char2comp.data(), sigma, comp2char.data());
  }
};

// A k-mer of the input graph as the reference's constructor and verifyIndex() see it: the label packed 3 bits per
// character above one byte of predecessor and one byte of successor comps (support.h:378-428).
typedef std::uint64_t key_type;

struct Key
{
  constexpr static size_type GCSA_CHAR_WIDTH = 3;
  constexpr static key_type  CHAR_MASK = 0x7;
  constexpr static size_type MAX_LENGTH = 16;
  constexpr static key_type  PRED_SUCC_MASK = 0xFFFF;

  static key_type encode(const Alphabet& alpha, const std::string& kmer, byte_type pred, byte_type succ)   // support.h:385-396
  {
    key_type packed = 0;
    for(char c : kmer) { packed = (packed << GCSA_CHAR_WIDTH) | alpha.char2comp[std::uint8_t(c)]; }
    return (((packed << 8) | pred) << 8) | succ;
  }
  static std::string decode(key_type key, size_type kmer_length, const Alphabet& alpha)                   // support.cpp:539-553
  {
    key_type chars = label(key);
    const size_type length = (kmer_length < MAX_LENGTH ? kmer_length : MAX_LENGTH);
    std::string result(length, '\0');
    for(size_type i = length; i > 0; i--) { result[i - 1] = char(alpha.comp2char[chars & CHAR_MASK]); chars >>= GCSA_CHAR_WIDTH; }
    return result;
  }
  static size_type label(key_type key) { return key >> 16; }
  static byte_type predecessors(key_type key) { return byte_type((key >> 8) & 0xFF); }
  static byte_type successors(key_type key) { return byte_type(key & 0xFF); }
  static comp_type last(key_type key) { return comp_type((key >> 16) & CHAR_MASK); }
};

struct KMer    // support.h:475-497
{
  key_type  key;
  node_type from, to;

  KMer() : key(0), from(0), to(0) {}
  KMer(key_type _key, node_type _from, node_type _to) : key(_key), from(_from), to(_to) {}
  bool operator<(const KMer& another) const { return Key::label(key) < Key::label(another.key); }
  bool sorted() const { return to == ~node_type(0); }
  void makeSorted() { to = ~node_type(0); }
};

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_SUPPORT_H
