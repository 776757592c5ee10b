// Host-side writer for the byte streams GCSA::serialize / LCPArray::serialize produce
// (reference src/gcsa.cpp:140-179, src/lcp.cpp:116-128), from the plain arrays of a gcsa2_host_view:
// the inverse of sdsl_reader.hpp, so that an index held by this engine can be handed back to tools
// built on the reference (GCSA::serialize of the facade, include/gcsa/gcsa.h).
//
// Member order and the GCSA-level headers are the reference's (cited per call in gcsa2_hip.hip).  The
// encodings of the SDSL containers are NOT in the reference tree; they follow sdsl-lite 2.1.1 as
// summarised in SURVEY.md section 8(f)-1: int_vector (u64 bit length [+ u8 width] + words),
// bit_vector_il<512> (one cumulative count before every 8 payload words, breadth-first rank samples),
// sd_vector (low / high parts + select_support_mcl<1>, <0> over the high part), select_support_mcl
// (4096-argument superblocks; "long" ones store every position, "mini" ones every 64th offset).
// FORMAT PARITY UNPINNED, exactly as for the reader: no file written by the real library exists here.

#ifndef GCSA2_SDSL_WRITER_HPP
#define GCSA2_SDSL_WRITER_HPP

#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace sdsl_file {

typedef void (*sink_fn)(void* ctx, const void* data, uint64_t bytes);

class Writer
{
public:
  Writer(sink_fn fn, void* ctx) : fn(fn), ctx(ctx), total(0) {}
  void raw(const void* data, uint64_t bytes) { if(bytes > 0) { fn(ctx, data, bytes); total += bytes; } }
  template<class T> void put(T value) { raw(&value, sizeof(T)); }
  uint64_t written() const { return total; }

  // sdsl::int_vector<64>
  void int_vector64(const uint64_t* values, uint64_t count)
  {
    put<uint64_t>(count * 64);
    raw(values, count * 8);
  }
  // sdsl::int_vector<8> / int_vector<0> of width 8 from bytes
  void int_vector8(const uint8_t* values, uint64_t count, bool with_width)
  {
    put<uint64_t>(count * 8);
    if(with_width) { put<uint8_t>(8); }
    raw(values, count);
    const uint64_t pad = (8 - count % 8) % 8, zero = 0;
    raw(&zero, pad);
  }
  // sdsl::int_vector<0>: packed words given
  void int_vector0_words(const uint64_t* words, uint64_t count, uint8_t width)
  {
    put<uint64_t>(count * width);
    put<uint8_t>(width);
    masked_words(words, count * width);
  }
  // sdsl::int_vector<0> from values
  void int_vector0(const std::vector<uint64_t>& values, uint8_t width)
  {
    std::vector<uint64_t> words((values.size() * width + 63) / 64 + 1, 0);
    for(uint64_t i = 0; i < values.size(); i++)
    {
      const uint64_t bit = i * width, w = bit >> 6, off = bit & 63;
      const uint64_t v = (width == 64 ? values[i] : values[i] & ((uint64_t(1) << width) - 1));
      words[w] |= v << off;
      if(off + width > 64) { words[w + 1] |= v >> (64 - off); }
    }
    int_vector0_words(words.data(), values.size(), width);
  }
  // sdsl::bit_vector
  void bit_vector(const uint64_t* words, uint64_t bits)
  {
    put<uint64_t>(bits);
    masked_words(words, bits);
  }

private:
  void masked_words(const uint64_t* words, uint64_t bits)     // ceil(bits / 64) words, padding bits cleared
  {
    const uint64_t full = bits / 64;
    raw(words, full * 8);
    if(bits & 63) { put<uint64_t>(words[full] & ((uint64_t(1) << (bits & 63)) - 1)); }
  }
  sink_fn fn; void* ctx; uint64_t total;
};

inline uint32_t bits_hi(uint64_t x) { return x == 0 ? 0 : 63 - uint32_t(__builtin_clzll(x)); }     // sdsl::bits::hi

inline uint64_t plain_word(const uint64_t* words, uint64_t bits, uint64_t i)    // word i of a vector of `bits` bits, zero padded
{
  const uint64_t count = (bits + 63) / 64;
  if(i >= count) { return 0; }
  uint64_t w = words[i];
  if(i == bits / 64 && (bits & 63)) { w &= (uint64_t(1) << (bits & 63)) - 1; }
  return w;
}

// sdsl::bit_vector_il<512>: size, block_num, superblocks, block_shift, data, rank_samples; then an empty rank support
inline void write_bit_vector_il(Writer& out, const uint64_t* words, uint64_t bits)
{
  const uint64_t payload = (bits + 64) / 64, superblocks = (bits + 512) / 512, mem = payload + superblocks + 1;
  std::vector<uint64_t> data(mem, 0);
  uint64_t cumulative = 0;
  for(uint64_t i = 0; i < payload; i++)
  {
    if((i & 7) == 0) { data[i + i / 8] = cumulative; }
    const uint64_t w = plain_word(words, bits, i);
    data[i + i / 8 + 1] = w;
    cumulative += uint64_t(__builtin_popcountll(w));
  }
  if((payload & 7) == 0 && payload / 8 < superblocks) { data[payload + payload / 8] = cumulative; }
  data[mem - 1] = cumulative;
  std::vector<uint64_t> samples;
  if(mem > 1024 * 64)                         // init_rank_samples: breadth-first midpoints of the binary search
  {
    const uint64_t want = (superblocks < 1024 * 64 ? superblocks : 1024 * 64);
    std::vector<std::pair<uint64_t, uint64_t>> queue;
    queue.emplace_back(0, superblocks);
    for(uint64_t head = 0; head < queue.size() && samples.size() < want; head++)
    {
      const uint64_t lb = queue[head].first, rb = queue[head].second, mid = lb + (rb - lb) / 2;
      samples.push_back(mid * 9 < mem ? data[mid * 9] : 0);
      queue.emplace_back(lb, mid);
      queue.emplace_back(mid + 1, rb);
    }
  }
  out.put<uint64_t>(bits); out.put<uint64_t>(mem); out.put<uint64_t>(superblocks); out.put<uint64_t>(9);
  out.int_vector64(data.data(), data.size());
  out.int_vector64(samples.data(), samples.size());
}

inline void write_empty_bit_vector_il(Writer& out)
{
  for(int i = 0; i < 4; i++) { out.put<uint64_t>(0); }
  out.int_vector64(nullptr, 0); out.int_vector64(nullptr, 0);
}

// sdsl::select_support_mcl<bit, 1> over a bit_vector of `bits` bits
inline void write_select_mcl(Writer& out, const uint64_t* words, uint64_t bits, bool bit)
{
  std::vector<uint64_t> args;
  for(uint64_t w = 0; w < (bits + 63) / 64; w++)
  {
    uint64_t word = plain_word(words, bits, w);
    if(!bit)
    {
      word = ~word;
      if(w == bits / 64 && (bits & 63)) { word &= (uint64_t(1) << (bits & 63)) - 1; }
    }
    while(word != 0) { args.push_back(64 * w + uint64_t(__builtin_ctzll(word))); word &= word - 1; }
  }
  out.put<uint64_t>(args.size());
  if(args.empty()) { return; }
  const uint64_t capacity = ((bits + 63) / 64) * 64;
  const uint32_t logn = bits_hi(capacity) + 1;
  const uint64_t logn4 = uint64_t(logn) * logn * logn * logn;
  const uint64_t sb = (args.size() + 4095) / 4096;
  std::vector<uint64_t> firsts(sb);
  for(uint64_t s = 0; s < sb; s++) { firsts[s] = args[4096 * s]; }
  out.int_vector0(firsts, uint8_t(logn));
  std::vector<uint64_t> kinds((sb + 63) / 64 + 1, 0);
  bool any_long = false;
  for(uint64_t s = 0; s < sb; s++)
  {
    const uint64_t a = 4096 * s, b = (a + 4096 < args.size() ? a + 4096 : args.size());
    if(args[b - 1] - args[a] > logn4) { any_long = true; } else { kinds[s >> 6] |= uint64_t(1) << (s & 63); }
  }
  if(any_long) { out.bit_vector(kinds.data(), sb); } else { out.bit_vector(kinds.data(), 0); }
  for(uint64_t s = 0; s < sb; s++)
  {
    const uint64_t a = 4096 * s, b = (a + 4096 < args.size() ? a + 4096 : args.size());
    const uint64_t diff = args[b - 1] - args[a];
    if(diff > logn4)                          // long superblock: every position, absolute
    {
      std::vector<uint64_t> full(4096, 0);
      for(uint64_t i = a; i < b; i++) { full[i - a] = args[i]; }
      out.int_vector0(full, uint8_t(bits_hi(args[b - 1]) + 1));
    }
    else                                      // mini blocks: every 64th position, relative to the first
    {
      std::vector<uint64_t> mini(64, 0);
      for(uint64_t i = a, j = 0; i < b; i += 64, j++) { mini[j] = args[i] - args[a]; }
      out.int_vector0(mini, uint8_t(bits_hi(diff) + 1));
    }
  }
}

// sdsl::sd_vector<>: size, wl, low, high, select_1 and select_0 over high (rank / select supports of the sd_vector
// itself are empty on disk)
inline void write_sd_vector(Writer& out, const uint64_t* words, uint64_t bits)
{
  std::vector<uint64_t> pos;
  for(uint64_t w = 0; w < (bits + 63) / 64; w++)
  {
    uint64_t word = plain_word(words, bits, w);
    while(word != 0) { pos.push_back(64 * w + uint64_t(__builtin_ctzll(word))); word &= word - 1; }
  }
  const uint64_t m = pos.size();
  uint32_t logm = bits_hi(m) + 1;
  const uint32_t logn = bits_hi(bits) + 1;
  if(logm == logn) { logm--; }
  const uint8_t wl = uint8_t(logn - logm);
  std::vector<uint64_t> low(m);
  const uint64_t high_len = m + (uint64_t(1) << logm);
  std::vector<uint64_t> high((high_len + 63) / 64 + 1, 0);
  for(uint64_t i = 0; i < m; i++)
  {
    low[i] = pos[i] & ((uint64_t(1) << wl) - 1);
    const uint64_t h = (pos[i] >> wl) + i;
    high[h >> 6] |= uint64_t(1) << (h & 63);
  }
  out.put<uint64_t>(bits); out.put<uint8_t>(wl);
  out.int_vector0(low, wl);
  out.bit_vector(high.data(), high_len);
  write_select_mcl(out, high.data(), high_len, true);
  write_select_mcl(out, high.data(), high_len, false);
}

inline void write_empty_sd_vector(Writer& out)
{
  out.put<uint64_t>(0); out.put<uint8_t>(0);
  out.put<uint64_t>(0); out.put<uint8_t>(64);       // int_vector<0>: empty, default width
  out.put<uint64_t>(0);                             // bit_vector: empty
  out.put<uint64_t>(0); out.put<uint64_t>(0);       // two empty select supports
}

} // namespace sdsl_file

#endif // GCSA2_SDSL_WRITER_HPP
