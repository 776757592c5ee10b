/*
 * gcsa_oracle.h -- CPU restatement of the GCSA2 query path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker the HIP engine is compared with (tests/, __graft_entry__.smoke()) and the
 * `cpu_baseline` leg of bench.py.  Nothing in gcsa2_amd/ may include, link or call it.
 *
 * Parity pin: the reference (jltsiren/gcsa2 @ v1.3.0) cannot be built in this image -- its one
 * external include <sdsl/wavelet_trees.hpp> (include/gcsa/utils.h:36, vgteam/sdsl-lite, no
 * pinned version, Makefile:1-2) is absent and un-vendored -- and the tree holds no golden
 * vectors.  The known-answer material is the paper's two worked examples: the GCSA of Figures 2-3
 * (paper/gcsa2_graph_dbg.ipe, paper/gcsa2_pruned_index.ipe -> tests/golden/paper_example.json) and the
 * text index of Figure 1 (paper/gcsa2_text_indexes.ipe: BWT, SA, LCP and LF columns of GCATCATA$ ->
 * tests/golden/text_example.json); this oracle is pinned against both and against a definition-level
 * brute force over the input graph (tests/naive.py).  SDSL's rank/select/access are unambiguous integer functions, so any
 * correct bitvector yields bit-identical range_type / node_type results.
 *
 * Memory layout mirrors what libgcsa2 + SDSL touch, so that the CPU baseline has the reference's
 * cache behaviour: bit_vector_il<512> = 1 cumulative word + 8 payload words (72-byte stride),
 * Elias-Fano for sparse_bwt and the Sadakane vectors, bit-packed stored_samples, byte LCP +
 * b-ary range-minimum tree.
 */
#ifndef GCSA_ORACLE_H
#define GCSA_ORACLE_H

#include <stdint.h>
#include "../include/gcsa2_hip.h"   /* gcsa2_host_view, gcsa2_stnode: the shared input description */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index;

oracle_index* oracle_create(const gcsa2_host_view* view);
void oracle_destroy(oracle_index* ix);
void oracle_free(void* p);

/* GCSA::find (include/gcsa/gcsa.h:96-110). */
void oracle_find(const oracle_index* ix, const uint8_t* pattern, uint64_t length,
                 uint64_t* sp, uint64_t* ep);
/* GCSA::charRange (gcsa.h:150-153). */
void oracle_char_range(const oracle_index* ix, uint8_t comp, uint64_t* sp, uint64_t* ep);
/* GCSA::LF(range, comp) (gcsa.h:155-162). */
void oracle_lf_range(const oracle_index* ix, uint64_t* sp, uint64_t* ep, uint8_t comp);
/* GCSA::LF(path_node) (gcsa.h:165-183). */
uint64_t oracle_lf_node(const oracle_index* ix, uint64_t node);
/* GCSA::LF_fast / LF_all (src/gcsa.cpp:742-798); out = sigma ranges, untouched entries (1,0). */
void oracle_lf_all(const oracle_index* ix, uint64_t sp, uint64_t ep, int all, uint64_t* out);
/* GCSA::count (src/gcsa.cpp:802-809). */
uint64_t oracle_count(const oracle_index* ix, uint64_t sp, uint64_t ep);
/* GCSA::locate(range, results, append=false, sort) (src/gcsa.cpp:827-842).
 * Returns a malloc'd array (oracle_free) of *count values. */
uint64_t* oracle_locate(const oracle_index* ix, uint64_t sp, uint64_t ep, int sort, uint64_t* count);
/* GCSA::locate(range, max_positions, results) (src/gcsa.cpp:844-878). */
uint64_t* oracle_locate_max(const oracle_index* ix, uint64_t sp, uint64_t ep, uint64_t max_positions,
                            uint64_t* count);
/* sampled / firstSample / lastSample / sample (gcsa.h:191-210). */
int oracle_sampled(const oracle_index* ix, uint64_t node);
uint64_t oracle_first_sample(const oracle_index* ix, uint64_t node);
int oracle_last_sample(const oracle_index* ix, uint64_t i);
uint64_t oracle_sample(const oracle_index* ix, uint64_t i);

/* LCPArray (include/gcsa/lcp.h:137-178, src/lcp.cpp:276-519). */
void oracle_parent(const oracle_index* ix, uint64_t sp, uint64_t ep, gcsa2_stnode* out);
uint64_t oracle_depth(const oracle_index* ix, uint64_t sp, uint64_t ep);
/* op: 0 psv, 1 psev, 2 nsv, 3 nsev */
void oracle_sv(const oracle_index* ix, int op, uint64_t pos, uint64_t* res_pos, uint64_t* res_val);
void oracle_rmq(const oracle_index* ix, uint64_t sp, uint64_t ep, uint64_t* res_pos, uint64_t* res_val);

/* Batched drivers: `threads` OpenMP threads, static contiguous split over queries
 * (the verifyIndex pattern, src/algorithms.cpp:106-114); threads = 1 is the query_gcsa-style
 * serial loop (benchmark/query_gcsa.cpp:92-97).  Return elapsed seconds (omp_get_wtime,
 * src/utils.cpp:131-135). */
double oracle_find_batch(const oracle_index* ix, const uint8_t* patterns, const uint64_t* offsets,
                         uint64_t nq, uint64_t* ranges, int threads);
double oracle_lf_batch(const oracle_index* ix, const uint64_t* in, const uint8_t* comps, uint64_t nq,
                       uint64_t* out, int threads);
double oracle_count_batch(const oracle_index* ix, const uint64_t* ranges, uint64_t nq,
                          uint64_t* counts, int threads);
double oracle_parent_batch(const oracle_index* ix, const uint64_t* ranges, uint64_t nq,
                           gcsa2_stnode* out, int threads);
double oracle_depth_batch(const oracle_index* ix, const uint64_t* ranges, uint64_t nq,
                          uint64_t* out, int threads);
/* CSR locate: offsets[nq+1]; *values malloc'd (oracle_free). */
double oracle_locate_batch(const oracle_index* ix, const uint64_t* ranges, uint64_t nq,
                           uint64_t* offsets, uint64_t** values, int threads);

/* Roofline accounting: number of DISTINCT device rank blocks a find() touches,
 * for a device layout with `block_bits` payload bits per block; a step whose sp and ep+1 probes
 * fall into the same block counts once.  Also returns the number of executed LF steps. */
void oracle_find_traffic(const oracle_index* ix, const uint8_t* patterns, const uint64_t* offsets,
                         uint64_t nq, uint64_t block_bits, uint64_t* blocks_touched,
                         uint64_t* lf_steps);

/* countKMers (src/algorithms.cpp:387-421); seed_length = KMerSearchParameters::SEED_LENGTH = 5. */
uint64_t oracle_count_kmers(const oracle_index* ix, uint64_t k, int include_ns, int force, uint64_t seed_length, int threads);

/* Matching statistics by LF + parent (paper.tex:344): ms has one uint16 per pattern byte. */
void oracle_match_stats(const oracle_index* ix, const uint8_t* pattern, uint64_t len, uint16_t* ms,
                        uint64_t* sp, uint64_t* ep, uint64_t* fallbacks);
double oracle_match_stats_batch(const oracle_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq,
                                uint16_t* ms, uint64_t* ranges, uint64_t* fallbacks, int threads);

/* compareKMers (src/algorithms.cpp:534-616): result = { shared, left only, right only }. */
void oracle_compare_kmers(const oracle_index* left, const oracle_index* right, uint64_t k, int include_ns, int force,
                          uint64_t* result);
/* The same with the states of the unique k-mers (8 u64 each: left range, right range, k, kmer[3]), as
 * the reference dumps them to output.left / output.right (src/algorithms.cpp:425-457, 606-610);
 * free the arrays with oracle_free. */
void oracle_compare_kmers_records(const oracle_index* left, const oracle_index* right, uint64_t k, int include_ns, int force,
                                  uint64_t* result, uint64_t** left_records, uint64_t** right_records);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
