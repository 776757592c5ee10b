// Host-side reader for the byte streams GCSA::serialize / LCPArray::serialize produce
// (reference src/gcsa.cpp:140-216, src/lcp.cpp:116-143): turns a `.gcsa` (+ optional `.lcp`) file
// into the plain arrays of a gcsa2_host_view.  No SDSL code is used or needed: every SDSL member is
// decoded from its serialized bytes and every rank / select support is either empty on disk
// (rank_support_il, rank_support_sd, select_support_sd) or self-describing and skipped
// (select_support_mcl) -- the device image carries its own directories.
//
// FORMAT PARITY IS UNPINNED.  The order of the members is the reference's (cited per call below); the
// byte encodings of the SDSL containers are NOT in the reference tree and no `.gcsa` file exists in
// this environment, so they follow sdsl-lite 2.1.1 as documented in SURVEY.md section 8(f)-1.  The
// reader is therefore strict: it checks every internal consistency condition the encodings imply
// (sizes, block counts, number of ones, monotonicity, exact end of file) and refuses a file it does not
// fully account for, instead of guessing.  Padding bits past the end of a vector are masked, not judged.

#ifndef GCSA2_SDSL_READER_HPP
#define GCSA2_SDSL_READER_HPP

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace sdsl_file {

typedef uint64_t u64;

struct FormatError : std::runtime_error { explicit FormatError(const std::string& what) : std::runtime_error(what) {} };

// Read-only mapping of a whole file.
class Mapping
{
public:
  explicit Mapping(const char* path) : base(nullptr), bytes(0)
  {
    int fd = ::open(path, O_RDONLY);
    if(fd < 0) { throw FormatError(std::string("cannot open ") + path); }
    struct stat st;
    if(::fstat(fd, &st) != 0) { ::close(fd); throw FormatError(std::string("cannot stat ") + path); }
    bytes = u64(st.st_size);
    if(bytes > 0)
    {
      void* p = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
      if(p == MAP_FAILED) { ::close(fd); throw FormatError(std::string("cannot map ") + path); }
      base = static_cast<const uint8_t*>(p);
    }
    ::close(fd);
  }
  ~Mapping() { if(base != nullptr) { ::munmap(const_cast<uint8_t*>(base), bytes); } }
  Mapping(const Mapping&) = delete;
  Mapping& operator=(const Mapping&) = delete;

  const uint8_t* base;
  u64 bytes;
};

// Bounds-checked cursor; values are little-endian and unaligned (an int_vector<0> header is 9 bytes).
class Cursor
{
public:
  Cursor(const Mapping& m, const std::string& what) : base(m.base), bytes(m.bytes), pos(0), what(what) {}
  Cursor(const void* data, u64 size, const std::string& what) : base(static_cast<const uint8_t*>(data)), bytes(size), pos(0), what(what) {}
  u64 consumed() const { return pos; }

  template<class T> T get(const char* field)
  {
    need(sizeof(T), field);
    T value; std::memcpy(&value, base + pos, sizeof(T)); pos += sizeof(T);
    return value;
  }
  const uint8_t* take(u64 n, const char* field) { need(n, field); const uint8_t* p = base + pos; pos += n; return p; }
  bool at_end() const { return pos == bytes; }
  u64 remaining() const { return bytes - pos; }
  [[noreturn]] void error(const std::string& message) const
  {
    throw FormatError(what + ": " + message + " (at byte " + std::to_string(pos) + " of " + std::to_string(bytes) + ")");
  }

private:
  void need(u64 n, const char* field) const
  {
    if(n > bytes - pos) { error(std::string("truncated while reading ") + field); }
  }
  const uint8_t* base; u64 bytes, pos; std::string what;
};

// sdsl::int_vector<w>: u64 length in bits, [u8 width when w == 0], ceil(bits / 64) words.
struct IntVector
{
  u64 bits = 0; uint8_t width = 64; const uint8_t* data = nullptr;
  u64 size() const { return width == 0 ? 0 : bits / width; }
  u64 words() const { return (bits + 63) / 64; }
  u64 word(u64 i) const { u64 w; std::memcpy(&w, data + 8 * i, 8); return w; }
  u64 get(u64 i) const     // element i, LSB-first packed
  {
    u64 bit = i * width, w = bit >> 6, off = bit & 63;
    u64 value = word(w) >> off;
    if(off + width > 64) { value |= word(w + 1) << (64 - off); }
    return width == 64 ? value : value & ((u64(1) << width) - 1);
  }
  void copy_words(std::vector<u64>& out) const     // + 2 spare words: word-granular readers may overrun
  {
    out.assign(words() + 2, 0);
    if(words() > 0) { std::memcpy(out.data(), data, 8 * words()); }
  }
};

inline IntVector read_int_vector(Cursor& in, uint8_t fixed_width, const char* field)
{
  IntVector v;
  v.bits = in.get<u64>(field);
  v.width = (fixed_width == 0 ? in.get<uint8_t>(field) : fixed_width);
  if(v.bits == 0 && v.width == 0) { v.width = 64; }        // empty vector: the width carries no information
  if(v.width == 0 || v.width > 64) { in.error(std::string(field) + ": invalid integer width"); }
  if(v.bits > (u64(1) << 46)) { in.error(std::string(field) + ": implausible length"); }
  if(v.bits % v.width != 0) { in.error(std::string(field) + ": length is not a multiple of the width"); }
  v.data = in.take(8 * v.words(), field);
  return v;
}

inline u64 popcount_words(const IntVector& v)
{
  u64 total = 0;
  for(u64 i = 0; i < v.words(); i++) { total += u64(__builtin_popcountll(v.word(i))); }
  return total;
}

// sdsl::bit_vector_il<512>: u64 size, block_num, superblocks, block_shift; int_vector<64> data with one
// cumulative count before every 8 payload words and a final total; int_vector<64> rank_samples.
inline void read_bit_vector_il(Cursor& in, std::vector<u64>& plain, u64& size, const char* field)
{
  size = in.get<u64>(field);
  u64 block_num = in.get<u64>(field), superblocks = in.get<u64>(field), block_shift = in.get<u64>(field);
  IntVector data = read_int_vector(in, 64, field);
  read_int_vector(in, 64, field);                          // rank_samples: binary-search accelerator, rebuilt nowhere
  plain.assign((size + 63) / 64 + 2, 0);
  if(data.bits == 0)                                       // default-constructed (unused comp)
  {
    if(size != 0 || block_num != 0) { in.error(std::string(field) + ": empty data in a non-empty bit_vector_il"); }
    return;
  }
  const u64 payload = (size + 64) / 64;
  if(block_shift != 9 || superblocks != (size + 512) / 512 || block_num != payload + superblocks + 1 || data.size() != block_num)
  {
    in.error(std::string(field) + ": bit_vector_il<512> block structure does not match its size");
  }
  u64 cumulative = 0;
  for(u64 i = 0; i < payload; i++)
  {
    if((i & 7) == 0 && data.word(i + i / 8) != cumulative) { in.error(std::string(field) + ": interleaved rank count mismatch"); }
    u64 w = data.word(i + i / 8 + 1);
    cumulative += u64(__builtin_popcountll(w));
    if(i < (size + 63) / 64) { plain[i] = w; }       // a spare payload word past the end carries no bits of the vector
  }
  if(data.word(block_num - 1) != cumulative) { in.error(std::string(field) + ": final rank count mismatch"); }
  if((size & 63) != 0) { plain[size / 64] &= (u64(1) << (size & 63)) - 1; }     // padding bits are not part of the vector
}

// sdsl::select_support_mcl<b, 1>: u64 arg_cnt; if non-zero: int_vector<0> superblock, bit_vector
// mini_or_long, then one int_vector<0> per 4096 arguments.  Skipped: select is not stored in the view.
inline void skip_select_mcl(Cursor& in, const char* field)
{
  u64 arg_cnt = in.get<u64>(field);
  if(arg_cnt == 0) { return; }
  u64 sb = (arg_cnt + 4095) >> 12;
  IntVector superblock = read_int_vector(in, 0, field);
  if(superblock.size() != sb) { in.error(std::string(field) + ": select_support_mcl superblock count mismatch"); }
  IntVector mini_or_long = read_int_vector(in, 1, field);
  if(mini_or_long.bits != 0 && mini_or_long.bits != sb) { in.error(std::string(field) + ": select_support_mcl block-type vector mismatch"); }
  for(u64 i = 0; i < sb; i++) { read_int_vector(in, 0, field); }
}

// sdsl::bit_vector (+ nothing): plain words.
inline void read_bit_vector(Cursor& in, std::vector<u64>& plain, u64& size, const char* field)
{
  IntVector v = read_int_vector(in, 1, field);
  size = v.bits;
  v.copy_words(plain);
  if((size & 63) != 0) { plain[size / 64] &= (u64(1) << (size & 63)) - 1; }       // padding bits are not part of the vector
}

// sdsl::sd_vector<>: u64 size, u8 wl, int_vector<0> low, bit_vector high, select_support_mcl<1>,
// select_support_mcl<0>.  One i (0-based) sits at ((select1(high, i) - i) << wl) | low[i].
inline void read_sd_vector(Cursor& in, std::vector<u64>& plain, u64& size, const char* field)
{
  size = in.get<u64>(field);
  uint8_t wl = in.get<uint8_t>(field);
  IntVector low = read_int_vector(in, 0, field);
  IntVector high = read_int_vector(in, 1, field);
  skip_select_mcl(in, field);
  skip_select_mcl(in, field);
  if(size > (u64(1) << 46)) { in.error(std::string(field) + ": implausible sd_vector size"); }
  plain.assign((size + 63) / 64 + 2, 0);
  const u64 ones = (low.bits == 0 ? 0 : low.size());
  if(ones > 0 && (low.width != wl || wl >= 64)) { in.error(std::string(field) + ": sd_vector low width mismatch"); }
  u64 seen = 0, previous = 0;
  for(u64 w = 0; w < high.words(); w++)
  {
    u64 word = high.word(w);
    while(word != 0)
    {
      u64 p = 64 * w + u64(__builtin_ctzll(word));
      word &= word - 1;
      if(seen >= ones) { in.error(std::string(field) + ": sd_vector has more high bits than low parts"); }
      u64 value = ((p - seen) << wl) | low.get(seen);
      if(value >= size || (seen > 0 && value <= previous)) { in.error(std::string(field) + ": sd_vector positions not increasing or out of range"); }
      plain[value >> 6] |= u64(1) << (value & 63);
      previous = value; seen++;
    }
  }
  if(seen != ones) { in.error(std::string(field) + ": sd_vector has fewer high bits than low parts"); }
}

} // namespace sdsl_file

#endif // GCSA2_SDSL_READER_HPP
