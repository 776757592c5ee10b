/*
 * gcsa2_hip.h -- C ABI of the MI355X batched backward-search engine for GCSA2 indexes.
 *
 * This is the drop-in boundary for the query hot path of jltsiren/gcsa2: every entry point
 * below replaces a method of `gcsa::GCSA` / `gcsa::LCPArray` that the reference implements as
 * header-inline C++ over SDSL bitvectors (citations are relative to the reference tree).
 * The reference has no FFI of its own -- its boundary is a C++ class API -- so this header is
 * what a C++ facade (include/gcsa2_hip/gcsa.hpp), a cgo/JNI stub or the ctypes binding in
 * gcsa2_amd/binding.py binds.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - All index data are described by plain LSB-first bit arrays (bit i of a vector lives in
 *     word i >> 6 at position i & 63, as in SDSL) and plain integer arrays.  The engine builds
 *     its own device image from them; nothing is kept by reference after create() returns.
 *   - Every bit array must be readable for ceil(bits / 64) whole words.
 *   - Ranges are closed [sp, ep] pairs of uint64 as in `gcsa::range_type`
 *     (include/gcsa/utils.h:84); a range is empty iff sp + 1 > ep + 1 (utils.h:93-96).
 *   - `*_batch` entry points take HOST pointers, copy in, run the kernels, copy out and
 *     synchronise.  `*_device` entry points take DEVICE pointers on the handle's device, only
 *     enqueue work on `stream` (a hipStream_t passed as void*, NULL = default stream) and do
 *     not synchronise -- they are what the benchmark and the multi-GPU driver use.
 *   - Device buffers of the `*_device` entry points: d_patterns is read in aligned 8-byte words, so the
 *     allocation must extend to the 8-byte boundary at or after its last pattern byte and start at or before the
 *     8-byte boundary at or before its first one (any hipMalloc'd buffer of `total bytes + 8` starting with the
 *     first pattern satisfies both); d_offsets, d_ranges and every other u64 array must be 8-byte aligned, range
 *     pairs 16-byte aligned; d_ms of gcsa2_match_stats_device must be 8-byte aligned and hold 4 spare entries.
 *     The host-pointer entry points have no such requirements (they stage into padded buffers).
 *   - Return value: GCSA2_OK (0) or a negative gcsa2_status.  Query entry points do not
 *     validate ranges or comps, exactly like the reference's low-level interface
 *     (include/gcsa/gcsa.h:133-135); count/locate apply the reference's `ep >= size()` guard.
 *   - A handle is immutable after creation and may be used from several host threads.
 */
#ifndef GCSA2_HIP_H
#define GCSA2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gcsa2_status {
  GCSA2_OK = 0,
  GCSA2_ERR_INVALID_ARGUMENT = -1,
  GCSA2_ERR_NO_DEVICE = -2,     /* no HIP device / runtime failure at init */
  GCSA2_ERR_OUT_OF_MEMORY = -3,
  GCSA2_ERR_HIP = -4,           /* any other HIP runtime error; see gcsa2_last_error() */
  GCSA2_ERR_MISSING_COMPONENT = -5, /* e.g. count() without counters, parent() without LCP */
  GCSA2_ERR_BUFFER_TOO_SMALL = -6
} gcsa2_status;

/*
 * Host-side description of one index = the data members of gcsa::GCSA
 * (include/gcsa/gcsa.h:214-240) and gcsa::LCPArray (include/gcsa/lcp.h:188-190) as plain arrays.
 */
typedef struct gcsa2_host_view {
  /* GCSAHeader (include/gcsa/files.h:135-156) */
  uint64_t path_nodes;            /* header.path_nodes = size() */
  uint64_t edges;                 /* header.edges */
  uint64_t order;                 /* header.order */

  /* Alphabet (include/gcsa/support.h:93-155) */
  uint64_t sigma;                 /* alpha.sigma, 1..GCSA2_MAX_SIGMA */
  uint64_t fast_chars;            /* alpha.fast_chars (informational: every comp is stored densely) */
  const uint8_t*  char2comp;      /* alpha.char2comp, 256 entries */
  const uint64_t* C;              /* alpha.C, sigma + 1 entries */

  /* fast_bwt[1..fast_chars] / sparse_bwt[0, fast_chars+1..] as plain bits: bwt[c], path_nodes bits */
  const uint64_t* const* bwt;     /* sigma pointers */
  const uint64_t* edge_bits;      /* edges, `edges` bits */
  const uint64_t* sampled_path_bits; /* sampled_paths, path_nodes bits; NULL = no locate support */

  /* stored_samples (sdsl::int_vector<0>) and samples (bit_vector) */
  uint64_t sample_count;          /* stored_samples.size() */
  uint64_t sample_width;          /* stored_samples.width(), 1..64 */
  const uint64_t* stored_samples; /* packed: element i = bits [i*w, (i+1)*w) */
  const uint64_t* sample_bits;    /* samples, sample_count bits */

  /* extra_pointers (SadaSparse, include/gcsa/support.h:298-364); NULL filter = no count support */
  const uint64_t* extra_filter_bits; /* filter, path_nodes bits */
  uint64_t extra_values_len;         /* values.size() */
  const uint64_t* extra_values_bits; /* values as plain bits */
  /* redundant_pointers (SadaCount, include/gcsa/support.h:231-279) */
  uint64_t redundant_len;            /* data.size() */
  const uint64_t* redundant_bits;    /* data */

  /* LCPArray (include/gcsa/lcp.h:182-190); lcp_data NULL = no parent/depth support */
  uint64_t lcp_size;              /* header.size = number of leaves */
  uint64_t lcp_branching;         /* header.branching */
  uint64_t lcp_levels;            /* offsets.size() - 1 */
  const uint64_t* lcp_offsets;    /* lcp_levels + 1 entries */
  const uint8_t*  lcp_data;       /* data widened to bytes, lcp_offsets[lcp_levels] entries */

  /* alpha.comp2char, sigma entries, or NULL (round 3; appended).  The query path never reads it; it is kept so that
   * load + serialize reproduces a file's alphabet.  NULL = derived from char2comp (gcsa2_derive_comp2char). */
  const uint8_t*  comp2char;
} gcsa2_host_view;

#define GCSA2_MAX_SIGMA 16

typedef struct gcsa2_index gcsa2_index;   /* opaque; owns the device image */

/* Suffix-tree node, field for field gcsa::STNode (include/gcsa/lcp.h:40-79). */
typedef struct gcsa2_stnode {
  uint64_t sp, ep;
  uint64_t left_lcp, right_lcp;
  uint64_t node_lcp;              /* GCSA2_UNKNOWN = STNode::UNKNOWN */
} gcsa2_stnode;

#define GCSA2_UNKNOWN (~(uint64_t)0)

/* ---- lifetime ----------------------------------------------------------------------------- */

/* Number of visible HIP devices, or a negative status. */
int gcsa2_device_count(void);

/* Build the device image of `view` on HIP device `device`.  Replaces GCSA::load + LCPArray::load
 * followed by nothing: the reference queries host RAM in place (benchmark/query_gcsa.cpp:55,63).
 * The image (rank blocks, select hints, fused LF blocks) is built ON the device: every bulk array of the view (bwt[c],
 * edge_bits, the sample and counter bit arrays, stored_samples, lcp_data) is copied to HBM once, 1/8 byte per bit, and
 * may itself be in host, pinned or device memory (hipMemcpyDefault).  char2comp, C, lcp_offsets, comp2char and the
 * pointer table `bwt` are read by the host.  No host copy of the image is made. */
int gcsa2_index_create(const gcsa2_host_view* view, int device, gcsa2_index** out);
void gcsa2_index_destroy(gcsa2_index* index);

/* The optional tables of an image -- memoisations of reference functions that change no result: the two-character pair
 * blocks (10.7 bytes per path node; two LF steps of gcsa.h:155-162 per memory request), the k-mer seed table (8 * 4^k bytes;
 * find() of every k-mer, gcsa.h:96-110) and the locate table (8 bytes per path node; the walk of locateInternal,
 * src/gcsa.cpp:880-896).  gcsa2_index_create builds what the free device memory allows, or what the environment says, read
 * once at create time: GCSA2_PAIR_BLOCKS=0, GCSA2_KMER_TABLE=k, GCSA2_LOCATE_TABLE=0, and GCSA2_MEMORY_BUDGET_MB=m, a cap on
 * the whole image under which the tables are taken in the order seed table (small) / pair blocks / seed table (grown) /
 * locate table.  gcsa2_index_set_tables re-shapes an existing image, e.g. to give memory back under pressure:
 * pair_blocks and locate_table: 0 = drop, 1 = build if absent, -1 = leave; kmer_k: 0 = drop, 1..16 = exactly that size,
 * -1 = leave.  It waits for the device to idle first; the caller must not run queries on this handle (or on facade copies
 * sharing it) during the call.  On failure the table in question is absent and every query still works. */
int gcsa2_index_set_tables(gcsa2_index* index, int pair_blocks, int kmer_k, int locate_table);
/* A handle keeps what its host-pointer entry points have grown to: the pipeline of gcsa2_find_batch (6 lanes x 2 sets of
 * ~14 MB: about 170 MB of page-locked host memory and as much device memory, made at the first batch of 2^19 or more
 * patterns), the staging objects of the small calls (8 MB pinned + a grow-only device arena each) and the stream-ordered
 * scratch pool of the queries.  gcsa2_index_trim gives all of it back (the next call builds what it needs again); like
 * set_tables it waits for the device and must not run beside queries on the same handle. */
int gcsa2_index_trim(gcsa2_index* index);
/* Shape of that pipeline: `lanes` host threads, each with its stream and two staging sets (1..16; default 6, or
 * GCSA2_PIPE_LANES at create time), 2^chunk_log2 patterns per chunk (15..20; default 18, GCSA2_PIPE_CHUNK); 0 leaves a value
 * as it is; blocking: 1 = the lanes sleep while they wait for a chunk (hipEventBlockingSync), 0 = they spin, -1 = as it is.
 * Trims the handle first (same rules as gcsa2_index_trim).  What is best depends on the host: tests/perf/
 * packed_pipeline.py sweeps both on a live image. */
int gcsa2_index_set_pipeline(gcsa2_index* index, int lanes, int chunk_log2, int blocking);

/* Thread-local description of the last failing call. */
const char* gcsa2_last_error(void);

/* GCSAHeader accessors (include/gcsa/gcsa.h:137-148). */
uint64_t gcsa2_size(const gcsa2_index* index);
uint64_t gcsa2_edge_count(const gcsa2_index* index);
uint64_t gcsa2_order(const gcsa2_index* index);
uint64_t gcsa2_sample_count(const gcsa2_index* index);
uint64_t gcsa2_sample_bits(const gcsa2_index* index);
int      gcsa2_device(const gcsa2_index* index);
/* Bytes of HBM held by the image. */
uint64_t gcsa2_device_bytes(const gcsa2_index* index);
/* Payload bits per rank block of the device layout (the unit of the roofline model). */
uint64_t gcsa2_block_bits(const gcsa2_index* index);

/* ---- find: GCSA::find(begin, end)  (include/gcsa/gcsa.h:96-122) ---------------------------- */

/* patterns = concatenated pattern bytes, pattern q = patterns[offsets[q] .. offsets[q+1]).
 * ranges[2q], ranges[2q+1] = (sp, ep), including the edge-space integers the reference returns
 * for a range that empties inside LF (gcsa.h:160).  */
int gcsa2_find_batch(const gcsa2_index* index, const uint8_t* patterns, const uint64_t* offsets,
                     uint64_t n_queries, uint64_t* ranges);
int gcsa2_find_device(const gcsa2_index* index, const uint8_t* d_patterns,
                      const uint64_t* d_offsets, uint64_t n_queries, uint64_t* d_ranges,
                      void* stream);

/* find() of n_queries patterns of ONE length (k-mer batches) handed over as 2-bit codes: 8 bytes per 32 characters over the link
 * instead of 32 pattern bytes + an 8-byte offset.  Pattern q occupies W = ceil(pattern_length / 32) u64 words at codes[q W ..];
 * word j holds the characters at distance 32 j .. 32 j + 31 from the pattern's END, the character at distance t in bits
 * [2 (t & 31), 2 (t & 31) + 2) as comp - 1 (A C G T = 0 1 2 3 with the default alphabet; unused high bits zero) -- the order
 * the backward search of gcsa.h:96-110 consumes them.  Only comps 1..4 can be written: a pattern with any other character
 * goes through gcsa2_find_batch.  Same ranges as the byte interface (the parity suite compares the two). */
int gcsa2_find_batch_packed(const gcsa2_index* index, const uint64_t* codes, uint64_t pattern_length, uint64_t n_queries, uint64_t* ranges);
int gcsa2_find_packed_device(const gcsa2_index* index, const uint64_t* d_codes, uint64_t pattern_length, uint64_t n_queries,
                             uint64_t* d_ranges, void* stream);

/* Launch shape.  variant 2 = the default (k_find2, fused 128-byte blocks, what
 * gcsa2_find_device runs); variant 4 = the same kernel behind a device-side sort of the queries by
 * pattern length, for batches of very uneven lengths (the 64 chains of a wavefront then finish
 * together; results are written in query order as always; stream-ordered scratch, no host sync).
 * Any other value is refused (rounds 1-2 also shipped a first-generation kernel and persistent
 * wavefronts as variants 1 and 5; neither won anywhere, both are gone). */
int gcsa2_find_device_variant(const gcsa2_index* index, int variant, const uint8_t* d_patterns,
                              const uint64_t* d_offsets, uint64_t n_queries, uint64_t* d_ranges,
                              void* stream);

/* Instrumented find for the roofline model (not the timed path): same results in d_ranges, and
 * d_stats[0] += number of distinct fused LF blocks fetched (gcsa2_find_block_bytes() each),
 * d_stats[1] += LF steps executed, d_stats[2] += seed-table lookups (8 bytes each), d_stats[3] += jump-table
 * lookups (16 bytes each).  A step whose two endpoints fall into one block counts once; a
 * two-character step counts as two LF steps and one block per distinct endpoint block, and a replayed pair
 * counts every block it fetched.  d_stats[4] += fetch rounds taken by lanes (one per single or pair step attempted),
 * d_stats[5] += those whose two endpoints lay in different blocks (a second fetch round for the whole wavefront),
 * d_stats[6] += seed-table entries that were marked "wide" (the range is then searched from scratch).
 * d_stats holds EIGHT words (rounds 1-2: four; an ABI change of round 3), zeroed by the caller.  d_stats[7] is an INPUT: 0, or
 * the device address of a bitmap of uint32 words, zeroed by the caller, with one bit per block of the image -- sigma x
 * (path_nodes / 384 + 1) FLB128 blocks, then 16 x (path_nodes / 192 + 1) FLP128 blocks; the kernel sets the bit of every
 * block it fetches, so that the set bits are the batch's working set in blocks (bench.py reports it). */
uint64_t gcsa2_find_block_bytes(const gcsa2_index* index);
/* Length k of the k-mer seed table built at create time (find() of every k-mer over comps 1..4,
 * memoised: a pattern whose last k characters are fast characters starts at step k).  0 = none.
 * One entry is 8 bytes (sp in 40 bits, range length in 24; the few ranges of 2^24 - 1 or more path nodes are
 * marked and searched from scratch).  Environment variable GCSA2_KMER_TABLE sets k (0 disables). */
uint64_t gcsa2_kmer_table_k(const gcsa2_index* index);
/* Bytes of the memoised locate table (0 = none): the walk of locateInternal (src/gcsa.cpp:880-896)
 * depends on the start node only, so it is run once per path node at create time and locate() reads
 * one 8-byte entry per path node instead of walking.  Needs samples; skipped when it would take
 * more than a third of the free device memory or when GCSA2_LOCATE_TABLE=0. */
uint64_t gcsa2_locate_table_bytes(const gcsa2_index* index);
/* Bytes of the jump table (0 = none; built when GCSA2_JUMP_TABLE=1): for every path node the chain of
 * up to 8 LF steps that is forced because each node on it has a single incoming label (a fast
 * character), with the node it ends in.  find() on a range of one path node whose next pattern
 * characters spell that chain moves there with one 16-byte lookup instead of one block fetch per
 * character; any other case steps as usual, so results (including the edge-space empty ranges of
 * include/gcsa/gcsa.h:160) are unchanged.  16 bytes per path node. */
uint64_t gcsa2_jump_table_bytes(const gcsa2_index* index);
/* Bytes of the two-characters-per-step blocks (0 = none).  For every ordered pair of fast characters the
 * composition of two LF steps (gcsa.h:155-162 applied twice) is stored as one rank structure of 128-byte
 * blocks over 192 path nodes each (gcsa2_amd/csrc/layout.hpp "FLP128"): find() then consumes two pattern
 * characters per memory request.  A block also tells which of its two steps empties, if any: an emptying second
 * step yields the edge-space integers of gcsa.h:160 directly, an emptying first step is replayed through the
 * single-character blocks, so every range is unchanged.  10.7 bytes per path node, built on the device at create
 * time; GCSA2_PAIR_BLOCKS=0 disables. */
uint64_t gcsa2_pair_block_bytes(const gcsa2_index* index);
int gcsa2_find_stats_device(const gcsa2_index* index, const uint8_t* d_patterns,
                            const uint64_t* d_offsets, uint64_t n_queries, uint64_t* d_ranges,
                            uint64_t* d_stats, void* stream);

/* ---- LF: GCSA::LF(range, comp) (gcsa.h:155-162), GCSA::LF(path_node) (gcsa.h:165-183),
 *      GCSA::charRange(comp) (gcsa.h:150-153) ------------------------------------------------- */
int gcsa2_lf_batch(const gcsa2_index* index, const uint64_t* ranges_in, const uint8_t* comps,
                   uint64_t n_queries, uint64_t* ranges_out);
int gcsa2_lf_device(const gcsa2_index* index, const uint64_t* d_ranges_in, const uint8_t* d_comps,
                    uint64_t n_queries, uint64_t* d_ranges_out, void* stream);
int gcsa2_lf_node_batch(const gcsa2_index* index, const uint64_t* nodes_in, uint64_t n_queries,
                        uint64_t* nodes_out);
int gcsa2_char_range(const gcsa2_index* index, uint8_t comp, uint64_t* sp, uint64_t* ep);
/* GCSA::LF_fast / LF_all (src/gcsa.cpp:742-798): ranges_out holds sigma ranges per query,
 * entries the reference leaves untouched are written as the empty range (1, 0).
 * all = 0: comps 1..fast_chars; all = 1: comps 1..sigma-2. */
int gcsa2_lf_all_batch(const gcsa2_index* index, const uint64_t* ranges_in, uint64_t n_queries,
                       int all, uint64_t* ranges_out);

/* ---- count: GCSA::count(range) (src/gcsa.cpp:802-809) -------------------------------------- */
int gcsa2_count_batch(const gcsa2_index* index, const uint64_t* ranges, uint64_t n_queries,
                      uint64_t* counts);
int gcsa2_count_device(const gcsa2_index* index, const uint64_t* d_ranges, uint64_t n_queries,
                       uint64_t* d_counts, void* stream);

/* ---- scalar calls: the per-character caller (include/gcsa/gcsa.h:155-162 in a loop; vg's MEM finder) ---------------------
 * A one-query call of gcsa2_lf_batch / gcsa2_lf_node_batch / gcsa2_count_batch / gcsa2_parent_batch -- what the facade's
 * scalar LF() / count() / parent() make -- does not launch a kernel: the request goes through a page-locked slot to ONE
 * resident wavefront of the index's device, which answers in the same slot (4 us per call instead of 12-17 through a launch;
 * a CPU LF step takes 0.35 us, paper.tex:408: batches remain the way to throughput).  The wavefront is launched by the first
 * such call, leaves by itself after GCSA2_MAILBOX_PARK_US (default 200) microseconds without a request -- that is how long a
 * device-wide synchronisation elsewhere in the process may wait for it -- and after GCSA2_MAILBOX_LIFE_MS (20) in any case; the
 * next call launches it again.  It occupies one wavefront slot of one CU while it lives.  One host thread at a time uses it;
 * a second thread's call takes the launch path meanwhile.  GCSA2_MAILBOX=0 (read at create time) switches it off.  Results
 * are those of the batch kernels bit for bit (the same device functions).  gcsa2_mailbox_stats: calls answered, launches. */
int gcsa2_mailbox_stats(const gcsa2_index* index, uint64_t* calls, uint64_t* launches);

/* ---- locate: GCSA::locate(range, results, append=false, sort) (src/gcsa.cpp:827-842) -------
 * Two calls, CSR output.  locate_run runs the whole query and keeps the result on the device
 * inside `*job`; locate_fetch copies the values (offsets[n_queries] of them) and frees the job.
 * sort != 0: sorted distinct values per query (removeDuplicates, utils.h:350-357), so
 *            offsets[q+1] - offsets[q] == count(range q).
 * sort == 0: values in path order, duplicates kept, exactly as the reference pushes them.
 * A batch of any size: one pass of the pipeline takes fewer than 2^31 values before deduplication (its library scans and
 * segmented sort count in int); a larger batch -- the paper's 16-mer batch has 2.5 G values -- is cut into consecutive
 * sub-batches of queries whose results are concatenated.  Only a single range with that many values is refused
 * (GCSA2_ERR_BUFFER_TOO_SMALL). */
typedef struct gcsa2_locate_job gcsa2_locate_job;
int gcsa2_locate_run(const gcsa2_index* index, const uint64_t* ranges, uint64_t n_queries,
                     int sort, uint64_t* offsets /* n_queries + 1 */, gcsa2_locate_job** job);
int gcsa2_locate_fetch(gcsa2_locate_job* job, uint64_t* values, uint64_t capacity);
void gcsa2_locate_discard(gcsa2_locate_job* job);
/* Device-resident form for pipelines and the benchmark: d_ranges in HBM; on return
 * *d_offsets / *d_values point into memory owned by the job (valid until discard). */
int gcsa2_locate_device(const gcsa2_index* index, const uint64_t* d_ranges, uint64_t n_queries,
                        int sort, gcsa2_locate_job** job, const uint64_t** d_offsets,
                        const uint64_t** d_values, uint64_t* total_values, void* stream);
/* The same query into caller-owned device buffers (no allocation of results, for pipelines that
 * reuse their buffers): d_offsets has n_queries + 1 entries, d_values room for `capacity` values.
 * *total_values = number of values; if that exceeds capacity the call fails with
 * GCSA2_ERR_BUFFER_TOO_SMALL (*total_values tells how much is needed) and nothing has been written
 * behind d_values + capacity -- the buffer itself may hold a prefix of the result by then: the values
 * are compacted into it before the host has seen their number.  On success d_values[0, *total_values)
 * is the result; what lies between *total_values and capacity is unspecified: a buffer with room for
 * the values BEFORE deduplication (sum of the ranges' value counts, >= *total_values) is used as the
 * sorts' work space and compacted in place, and when no range repeats a value -- a text index -- nothing
 * is compacted at all.  Complete on return. */
int gcsa2_locate_into(const gcsa2_index* index, const uint64_t* d_ranges, uint64_t n_queries, int sort,
                      uint64_t* d_offsets, uint64_t* d_values, uint64_t capacity,
                      uint64_t* total_values, void* stream);

/* GCSA::locate(range, max_positions, results) (src/gcsa.cpp:844-878): at most max_positions
 * distinct values, chosen with the reference's std::mt19937_64(sp ^ ep) draws, sorted.
 * *count = number of values written (or needed, with GCSA2_ERR_BUFFER_TOO_SMALL). */
int gcsa2_locate_max(const gcsa2_index* index, uint64_t sp, uint64_t ep, uint64_t max_positions,
                     uint64_t* values, uint64_t capacity, uint64_t* count);

/* sampled / sampleRange / firstSample (gcsa.h:191-206): out[3q] = sampled(node),
 * out[3q+1] = sampleRange(node).first (= firstSample(node)), out[3q+2] = sampleRange(node).second. */
int gcsa2_sample_range_batch(const gcsa2_index* index, const uint64_t* nodes, uint64_t n_queries,
                             uint64_t* out);
/* sample(i) and lastSample(i) (gcsa.h:208-210). */
int gcsa2_sample_batch(const gcsa2_index* index, const uint64_t* sample_indexes, uint64_t n_queries,
                       uint64_t* values, uint8_t* last_flags);
/* sampledPositions() (gcsa.h:143-148) and the Alphabet members callers read (support.h:150-151). */
uint64_t gcsa2_sampled_positions(const gcsa2_index* index);
uint64_t gcsa2_sigma(const gcsa2_index* index);
uint64_t gcsa2_fast_chars(const gcsa2_index* index);
void gcsa2_alphabet(const gcsa2_index* index, uint8_t* char2comp /* 256 */, uint64_t* C /* sigma + 1 */);
/* comp2char of an alphabet given by char2comp alone: per comp the first byte that is not a lower-case letter and not NUL
 * (else the first byte), and the reference's "$ACGTN#" for its default alphabet (src/support.cpp:69-92).  The one rule behind
 * the facade's Alphabet and GCSA::serialize when a view carries no comp2char.  Host only. */
void gcsa2_derive_comp2char(const uint8_t* char2comp /* 256 */, uint64_t sigma, uint8_t* comp2char /* sigma */);

/* ---- suffix-tree operations: LCPArray::parent / depth / psv / psev / nsv / nsev / rmq
 *      (include/gcsa/lcp.h:137-178, src/lcp.cpp:276-519) -------------------------------------- */
int gcsa2_parent_batch(const gcsa2_index* index, const uint64_t* ranges, uint64_t n_queries,
                       gcsa2_stnode* nodes);
int gcsa2_parent_device(const gcsa2_index* index, const uint64_t* d_ranges, uint64_t n_queries,
                        gcsa2_stnode* d_nodes, void* stream);
int gcsa2_depth_batch(const gcsa2_index* index, const uint64_t* ranges, uint64_t n_queries,
                      uint64_t* depths);
/* op: 0 psv, 1 psev, 2 nsv, 3 nsev.  results[2q], results[2q+1] = (position, LCP value) or
 * notFound() = (values(), values()). */
int gcsa2_sv_batch(const gcsa2_index* index, int op, const uint64_t* positions,
                   uint64_t n_queries, uint64_t* results);
/* rmq(sp, ep): leftmost minimum of LCP[sp..ep] as (position, value), or notFound(). */
int gcsa2_rmq_batch(const gcsa2_index* index, const uint64_t* ranges, uint64_t n_queries,
                    uint64_t* results);

/* LCPArray::size / values / levels / branching / operator[] (lcp.h:124-129). */
uint64_t gcsa2_lcp_size(const gcsa2_index* index);
uint64_t gcsa2_lcp_values(const gcsa2_index* index);
uint64_t gcsa2_lcp_levels(const gcsa2_index* index);
uint64_t gcsa2_lcp_branching(const gcsa2_index* index);
int gcsa2_lcp_access_batch(const gcsa2_index* index, const uint64_t* positions, uint64_t n_queries,
                           uint64_t* out);

/* ---- countKMers(index, k, parameters) (include/gcsa/algorithms.h:73-84, src/algorithms.cpp:387-421):
 * number of distinct k-mers in the index = non-empty states at depth k of the search tree,
 * expanding with LF_fast (bases only) or, include_ns != 0, LF_all.  k > order() yields 0 unless
 * force != 0, k == 0 yields 1, as in the reference. */
int gcsa2_count_kmers(const gcsa2_index* index, uint64_t k, int include_ns, int force, uint64_t* result);

/* compareKMers(left, right, k, parameters) (include/gcsa/algorithms.h:86-92, src/algorithms.cpp:534-616):
 * result[0..2] = number of k-mers in both indexes, only in the left, only in the right one.  Same early exits as the
 * reference (k == 0 -> {1,0,0}; k > order without force, k > 64, incompatible alphabets -> zeros).
 * Both indexes must be on the same device. */
int gcsa2_compare_kmers(const gcsa2_index* left, const gcsa2_index* right, uint64_t k, int include_ns,
                        int force, uint64_t* result);
/* The same, also returning the search states of the k-mers found in one index only, as the reference
 * writes them to parameters.output + ".left" / ".right" (src/algorithms.cpp:425-457, 606-610): 8 u64
 * per state = left range, right range, k, kmer[3] (3 bits per comp, extension step i at bits
 * [3i, 3i+3), i.e. the LAST character of the k-mer first).  Order within a buffer is unspecified (the
 * reference's depends on its OpenMP schedule).  Capacities are in states; if result[1] / result[2]
 * exceed them the call fails with GCSA2_ERR_BUFFER_TOO_SMALL and result[] holds the sizes needed. */
int gcsa2_compare_kmers_records(const gcsa2_index* left, const gcsa2_index* right, uint64_t k,
                                int include_ns, int force, uint64_t* result,
                                uint64_t* left_records, uint64_t left_capacity,
                                uint64_t* right_records, uint64_t right_capacity);

/* ---- matching statistics: the LF + parent interplay of vg's MEM finder, fused ---------------
 * (paper/paper.tex:344 "maximal exact matches by using LF-mapping and parent
 * queries").  Composition of GCSA::LF(range, comp) (gcsa.h:155-162) and LCPArray::parent(range)
 * (src/lcp.cpp:276-301), scanning each pattern right to left: on an empty LF result the range is
 * replaced by its parent and the character retried; at the root the character is skipped.
 * ms[offsets[q] + i] = length of the longest match starting at byte i of pattern q that the
 * index reports (capped at 65535; exact for lengths <= order()), ranges[2q..] = range of that
 * match for i = 0, fallbacks[q] = number of parent() calls (may be NULL).  Needs the LCP array. */
int gcsa2_match_stats_batch(const gcsa2_index* index, const uint8_t* patterns, const uint64_t* offsets,
                            uint64_t n_queries, uint16_t* ms, uint64_t* ranges, uint64_t* fallbacks);
int gcsa2_match_stats_device(const gcsa2_index* index, const uint8_t* d_patterns, const uint64_t* d_offsets,
                             uint64_t n_queries, uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks,
                             void* stream);
/* Launch shape, same results.  variant 2 = one lane per pattern (wave-cooperative block fetch); variant 5 = the same kernel
 * as persistent wavefronts whose idle lanes draw the next pattern from a counter -- for batches of ragged pattern lengths
 * or of uneven difficulty (a pattern with mismatches takes three times the rounds of one without); variant 0 = the
 * library chooses (what gcsa2_match_stats_device runs): persistent lanes when the batch is more than two generations of
 * resident workgroups, and gcsa2_match_stats_batch also when the longest pattern exceeds 1.25 x the mean.  Any other
 * value is refused.
 * The kernel reads the patterns as 2-bit codes prepared by a pre-pass into stream-ordered scratch, whose size depends on
 * d_offsets[n_queries]: gcsa2_match_stats_device and _variant read that value back (ONE wait for `stream` per call);
 * gcsa2_match_stats_device_sized takes it from the caller (total_pattern_bytes) and only enqueues. */
int gcsa2_match_stats_device_variant(const gcsa2_index* index, int variant, const uint8_t* d_patterns,
                                     const uint64_t* d_offsets, uint64_t n_queries, uint16_t* d_ms,
                                     uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream);
int gcsa2_match_stats_device_sized(const gcsa2_index* index, int variant, const uint8_t* d_patterns,
                                   const uint64_t* d_offsets, uint64_t n_queries, uint64_t total_pattern_bytes,
                                   uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream);
/* The same backward search with parent() on failure, reported as BREAK POINTS instead of one statistic per position: the
 * left-maximal matches a MEM finder collects (vg's caller shape; the reference exercises the LF + parent interplay in
 * verifyIndex, src/algorithms.cpp:146-167).  A record {position p, length, sp, ep} says that P[p, p + length) occurs with
 * path-node range (sp, ep) = find() of that substring (include/gcsa/gcsa.h:96-110) and cannot be extended by P[p - 1];
 * exactly the positions p with p == 0 or ms[p - 1] != ms[p] + 1 have a record (an empty pattern has none; a record of length 0
 * -- no character of the index at p, nor at p - 1 -- carries the whole index as its range).  Records of pattern q: d_breaks[d_break_offsets[q] .. d_break_offsets[q + 1]), in the order
 * of discovery (descending position).  The dense statistics follow from them: ms[i] = length - (i - p) for the record with the
 * largest p <= i.  d_break_offsets: n_queries + 1 entries; capacity: records d_breaks holds; *total_breaks: records found --
 * GCSA2_ERR_BUFFER_TOO_SMALL with that number when it exceeds capacity (nothing is written then).  d_ranges (final ranges) and
 * d_fallbacks (parent() calls per pattern) may be NULL.  variant as in gcsa2_match_stats_device_variant.  min_length > 0 keeps
 * only the records of at least that length -- a MEM finder's minimum match length: right after a mismatch the matches are as
 * short as any string of log4(n) characters and every position is a break point, which on a whole-genome index makes more
 * bytes of records than the dense statistics have; the dense form no longer follows from the records then.  Complete on return. */
typedef struct gcsa2_break { uint64_t position, length, sp, ep; } gcsa2_break;
int gcsa2_match_breaks_device(const gcsa2_index* index, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t n_queries,
                              uint64_t total_pattern_bytes, int variant, uint64_t min_length, uint64_t* d_break_offsets, gcsa2_break* d_breaks,
                              uint64_t capacity, uint64_t* total_breaks, uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream);
/* The same for a batch in host memory (offsets[0] == 0): copies in, runs, copies the CSR out.  ranges / fallbacks may be NULL.
 * A batch of two pieces' worth of pattern bytes or more goes in pieces (32 MB each, GCSA2_MS_PIECE_MB; four host threads with a stream each)
 * whose records are committed in pattern order; with GCSA2_ERR_BUFFER_TOO_SMALL (*total_breaks = the records of the whole batch)
 * the contents of the result arrays are unspecified. */
int gcsa2_match_breaks_batch(const gcsa2_index* index, const uint8_t* patterns, const uint64_t* offsets, uint64_t n_queries,
                             uint64_t min_length, uint64_t* break_offsets, gcsa2_break* breaks, uint64_t capacity,
                             uint64_t* total_breaks, uint64_t* ranges, uint64_t* fallbacks);
/* Diagnostic (not the timed path): the default kernel instrumented with shader-clock counters, same results.  d_prof[16],
 * zeroed by the caller: [0..7] cycles summed over the wavefronts for the phases of a round (loop head / pattern window, step
 * setup, first block fetch, first evaluation, second fetch + evaluation, outcome + statistics, parent() from the LCP chunks,
 * parent() tree walk); [8..15] events: rounds, rounds with a second fetch, lane steps, pair attempts, failed pair attempts,
 * parent() calls, tree walks, second fetches of lanes.  Needs the pair blocks. */
int gcsa2_match_stats_profile_device(const gcsa2_index* index, const uint8_t* d_patterns, const uint64_t* d_offsets,
                                     uint64_t n_queries, uint64_t total_pattern_bytes, uint16_t* d_ms, uint64_t* d_ranges,
                                     uint64_t* d_fallbacks, uint64_t* d_prof, void* stream);

/* ---- host-view container file ("G2HV") ---------------------------------------------------
 * Interchange between a process that can read .gcsa / .lcp files (the reference linked with SDSL:
 * GCSA::load, src/gcsa.cpp:184-216; LCPArray::load, src/lcp.cpp:130-143; exporter stub in
 * INTEGRATION.md) and GPU nodes that cannot.  The file is the gcsa2_host_view verbatim: header
 * fields, then every array with its byte length.  A bad tag or version fails like the reference's
 * load() does ("Invalid header", src/gcsa.cpp:188-193), as an error code.  These three functions
 * are host-only (no device needed). */
typedef struct gcsa2_view_storage gcsa2_view_storage;   /* owns the arrays of a loaded view */
int gcsa2_host_view_save(const gcsa2_host_view* view, const char* path);
int gcsa2_host_view_load(const char* path, gcsa2_view_storage** out);
const gcsa2_host_view* gcsa2_host_view_get(const gcsa2_view_storage* storage);
void gcsa2_host_view_free(gcsa2_view_storage* storage);
/* load + gcsa2_index_create in one call */
int gcsa2_index_create_from_file(const char* path, int device, gcsa2_index** out);

/* ---- the reference's own files ----------------------------------------------------------------
 * GCSA::load (src/gcsa.cpp:184-216) and LCPArray::load (src/lcp.cpp:130-143) for `.gcsa` / `.lcp`
 * files written by GCSA::serialize / LCPArray::serialize (GCSA version 3, LCP version 1), without
 * SDSL: each serialized SDSL container is decoded into the plain arrays of a gcsa2_host_view and the
 * rank / select supports are skipped.  lcp_path may be NULL (no parent / depth support then).
 * An invalid header fails like the reference's load() ("Invalid header", src/gcsa.cpp:188-193), as
 * GCSA2_ERR_INVALID_ARGUMENT; so does any structural inconsistency or any unaccounted byte.
 * FORMAT PARITY UNPINNED: the container encodings follow sdsl-lite 2.1.1 as recalled in SURVEY.md
 * section 8(f)-1; no real file was available to validate against (gcsa2_amd/csrc/sdsl_reader.hpp). */
int gcsa2_host_view_load_gcsa(const char* gcsa_path, const char* lcp_path, gcsa2_view_storage** out);
int gcsa2_index_create_from_gcsa(const char* gcsa_path, const char* lcp_path, int device, gcsa2_index** out);

/* The same from memory (what GCSA::load(std::istream&) / LCPArray::load(std::istream&) of the facade call after
 * reading the stream): parses one serialized structure at the start of `bytes`.  consumed != NULL: receives the
 * number of bytes the structure occupies, trailing bytes are left alone; consumed == NULL: the buffer must end
 * with the structure.  parse_lcp yields a view in which only the lcp_* fields are set. */
int gcsa2_host_view_parse_gcsa(const void* bytes, uint64_t size, uint64_t* consumed, gcsa2_view_storage** out);
int gcsa2_host_view_parse_lcp(const void* bytes, uint64_t size, uint64_t* consumed, gcsa2_view_storage** out);
/* GCSA::serialize (src/gcsa.cpp:140-179) / LCPArray::serialize (src/lcp.cpp:116-128) of a host view: the byte
 * stream is handed to `sink` piece by piece (an std::ostream::write in the facade).  Same caveat as for the
 * reader: the SDSL container encodings are restated from sdsl-lite 2.1.1, format parity is unpinned.
 * serialize_gcsa needs a view with samples and counters, serialize_lcp one with an LCP array. */
typedef void (*gcsa2_sink)(void* ctx, const void* data, uint64_t bytes);
int gcsa2_host_view_serialize_gcsa(const gcsa2_host_view* view, gcsa2_sink sink, void* ctx, uint64_t* written);
int gcsa2_host_view_serialize_lcp(const gcsa2_host_view* view, gcsa2_sink sink, void* ctx, uint64_t* written);
/* A device image holding only the LCP array of `view` (its GCSA fields are ignored): a stand-alone
 * gcsa::LCPArray as LCPArray::load (src/lcp.cpp:130-143) produces it.  Serves gcsa2_parent_* / depth / sv / rmq /
 * lcp_* ; the GCSA queries find nothing on it. */
int gcsa2_lcp_create(const gcsa2_host_view* view, int device, gcsa2_index** out);

/* ---- single-process multi-GPU -------------------------------------------------------------
 * A group holds one replica of the index per listed device (a device may be listed more than
 * once).  group_find_batch splits the batch into contiguous shards (sizes differ by at most one,
 * the static split of verifyIndex, src/algorithms.cpp:106-114), runs every shard on its device
 * from its own host thread and writes the ranges in query order.  Multi-process deployments
 * (one rank per GPU + one RCCL gather) use gcsa2_find_device from each rank instead. */
typedef struct gcsa2_group gcsa2_group;
int gcsa2_group_create(const gcsa2_host_view* view, const int* devices, int n_devices, gcsa2_group** out);
void gcsa2_group_destroy(gcsa2_group* group);
int gcsa2_group_size(const gcsa2_group* group);
const gcsa2_index* gcsa2_group_index(const gcsa2_group* group, int i);
int gcsa2_group_find_batch(const gcsa2_group* group, const uint8_t* patterns, const uint64_t* offsets,
                           uint64_t n_queries, uint64_t* ranges);

/* The same split with every shard already in the HBM of its device: d_patterns[r] / d_offsets[r] (offsets
 * rebased to 0, counts[r] + 1 entries) live on the device of replica r, d_ranges_root (2 x sum of counts words)
 * on the device of replica 0.  Every replica searches its shard on its own stream; the ranges are then gathered
 * in the root's HBM, in query order, by one grouped RCCL send / recv over xGMI (ncclCommInitAll over the device
 * list, created at the first call) -- or by peer copies when the group lists a device twice or RCCL is not
 * available (gcsa2_group_uses_rccl tells which).  Complete on return. */
int gcsa2_group_find_device(gcsa2_group* group, const uint8_t* const* d_patterns, const uint64_t* const* d_offsets,
                            const uint64_t* counts, uint64_t* d_ranges_root);
int gcsa2_group_uses_rccl(const gcsa2_group* group);

/* BASELINE configs[4] over the group -- the two queries of benchmark/query_gcsa.cpp:143-179 whose results are ragged or
 * long -- with the same contiguous split.  Both are complete on return.
 * group_match_stats_device: replica r computes the matching statistics of its shard (d_patterns[r], d_offsets[r] rebased
 *   to 0, counts[r] patterns of pattern_bytes[r] bytes in all); the statistics (sum of pattern_bytes entries + 4 spare,
 *   8-byte aligned), ranges and parent() counts (may be NULL) are gathered on replica 0's device in query order.
 * group_locate_device: replica r locates d_ranges[r] (counts[r] ranges); the per-replica totals come first, then the CSR
 *   offsets (rebased; d_offsets_root has sum of counts + 1 entries) and the values are gathered.  *job owns
 *   the values on replica 0's device (gcsa2_locate_discard). */
int gcsa2_group_match_stats_device(gcsa2_group* group, const uint8_t* const* d_patterns, const uint64_t* const* d_offsets,
                                   const uint64_t* counts, const uint64_t* pattern_bytes, uint16_t* d_ms_root,
                                   uint64_t* d_ranges_root, uint64_t* d_fallbacks_root);
int gcsa2_group_locate_device(gcsa2_group* group, const uint64_t* const* d_ranges, const uint64_t* counts, int sort,
                              uint64_t* d_offsets_root, gcsa2_locate_job** job, const uint64_t** d_values_root,
                              uint64_t* total_values);

/* ---- multi-process multi-GPU: one rank per GPU, one gather of hit ranges ----------------------
 * The single collective of the path: every rank searches its shard with gcsa2_find_device and
 * the (sp, ep) pairs are gathered in the root's HBM with grouped ncclSend / ncclRecv -- every peer uses its own
 * xGMI link into the root.  The communicator is RCCL's (ncclCommInitRank); the 128-byte id is created on one
 * rank and handed to the others by whatever launcher the application has (bench.py: the torch.distributed store).
 * RCCL is bound at run time; without it these calls fail with GCSA2_ERR_MISSING_COMPONENT. */
#define GCSA2_COMM_ID_BYTES 128
typedef struct gcsa2_comm gcsa2_comm;
int gcsa2_comm_unique_id(uint8_t* id /* GCSA2_COMM_ID_BYTES */);
int gcsa2_comm_create(const uint8_t* id, int rank, int world, int device, gcsa2_comm** out);
/* A communicator over the APPLICATION's transport instead of RCCL (MPI, a gather through host memory on hosts without RCCL,
 * the world-size-2 tests of this repository on one GPU): `gather` has the contract of gcsa2_comm_gather below -- rank r
 * contributes bytes[r] bytes of device memory at d_send, the root receives all parts back to back, in rank order, in device
 * memory at d_recv (its own part included) -- and is called by gcsa2_comm_gather / _match_stats / _locate wherever they would
 * use RCCL.  It may work asynchronously on `stream` or complete before it returns (after waiting for `stream`, on which the
 * data it sends was produced); 0 = success.  `user` is passed through.  gcsa2_comm_rccl_ranks reports 0 for such a
 * communicator.  d_send is always a valid device address, also when bytes[rank] == 0 (nothing is to be read from it then). */
typedef int (*gcsa2_gather_fn)(void* user, const void* d_send, const uint64_t* bytes, void* d_recv, int root, void* stream);
int gcsa2_comm_create_custom(int rank, int world, int device, gcsa2_gather_fn gather, void* user, gcsa2_comm** out);
void gcsa2_comm_destroy(gcsa2_comm* comm);
int gcsa2_comm_rank(const gcsa2_comm* comm);
int gcsa2_comm_world(const gcsa2_comm* comm);
/* The number of ranks RCCL itself reports for the communicator (ncclCommCount): what a benchmark prints to show that the
 * gather really spanned `world` devices. */
int gcsa2_comm_rccl_ranks(const gcsa2_comm* comm, int* ranks);
/* Rank r contributes bytes[r] bytes from d_send (bytes[] has `world` entries and is the same on every rank);
 * the root receives them back to back in rank order in d_recv (its own part by a device copy; d_recv may be
 * NULL elsewhere).  Enqueues on `stream` of the communicator's device and does not synchronise. */
int gcsa2_comm_gather(gcsa2_comm* comm, const void* d_send, const uint64_t* bytes, void* d_recv, int root, void* stream);
/* The same two queries with one rank per GPU: every rank passes its shard and the per-rank sizes (counts[r] queries and, for
 * the matching statistics, pattern_bytes[r] pattern bytes of rank r: the same arrays on every rank); the results land on
 * `root` in query order through gcsa2_comm_gather's grouped send / recv.  The *_root arguments are read on the root only.
 * comm_match_stats only enqueues on `stream` (all three result arrays are required on the root); comm_locate is complete on
 * return: the per-rank totals travel first, the root sizes the value buffer from them ("all-gather of
 * per-rank counts, then gatherv of the CSR values"). */
int gcsa2_comm_match_stats(gcsa2_comm* comm, const gcsa2_index* index, const uint8_t* d_patterns, const uint64_t* d_offsets,
                           const uint64_t* counts, const uint64_t* pattern_bytes, int root, uint16_t* d_ms_root,
                           uint64_t* d_ranges_root, uint64_t* d_fallbacks_root, void* stream);
int gcsa2_comm_locate(gcsa2_comm* comm, const gcsa2_index* index, const uint64_t* d_ranges, const uint64_t* counts, int sort,
                      int root, uint64_t* d_offsets_root, gcsa2_locate_job** job_root, const uint64_t** d_values_root,
                      uint64_t* total_values, void* stream);
/* Wire format for indexes whose path node and edge numbers are all below 2^32: (sp, ep) u64 pairs <->
 * (sp, ep + 1 - sp) u32 pairs, exact for every range find() returns (an empty range is (x, x - 1),
 * include/gcsa/utils.h:93-96).  Halves the bytes of the gather.  Launch on the current device. */
int gcsa2_pack_ranges32_device(const uint64_t* d_ranges, uint64_t n_queries, uint32_t* d_packed, void* stream);
int gcsa2_unpack_ranges32_device(const uint32_t* d_packed, uint64_t n_queries, uint64_t* d_ranges, void* stream);
/* The same for indexes whose path node and edge numbers are below 2^40 (BASELINE configs[3]: 5.7 G path nodes): 10 bytes per
 * range (sp and length, 40 bits each) instead of 16; d_packed holds 10 * n_queries bytes, 2-byte aligned. */
int gcsa2_pack_ranges40_device(const uint64_t* d_ranges, uint64_t n_queries, void* d_packed, void* stream);
int gcsa2_unpack_ranges40_device(const void* d_packed, uint64_t n_queries, uint64_t* d_ranges, void* stream);
/* Six bytes per range for the common case (same indexes): sp in 40 bits, the length in one byte, and behind the shard's ranges
 * a list of (query, length) pairs, `capacity` of them at most, for the ranges of 255 and more path nodes.  A shard's block has
 * gcsa2_wire48_bytes(n_queries, capacity) bytes whatever the ranges are -- the gather (src/algorithms.cpp:106-114 is the static
 * split it serves) keeps fixed sizes --, 16-byte aligned.  *d_overflow_count (optional, device memory) receives the number of
 * long ranges the shard had: a value beyond `capacity` means the block does not hold the batch (the ranges that did not fit
 * come out 255 path nodes long) and the caller must use the 40-bit pairs; nothing is truncated silently. */
uint64_t gcsa2_wire48_bytes(uint64_t n_queries, uint64_t capacity);
int gcsa2_pack_ranges48_device(const uint64_t* d_ranges, uint64_t n_queries, void* d_packed, uint64_t capacity, void* stream);
int gcsa2_unpack_ranges48_device(const void* d_packed, uint64_t n_queries, uint64_t capacity, uint64_t* d_ranges,
                                 uint64_t* d_overflow_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCSA2_HIP_H */
