/*
 * gcsa_oracle.c -- CPU restatement of the GCSA2 query path.  TEST INFRASTRUCTURE ONLY
 * (see gcsa_oracle.h for the parity-pin statement).  Plain C11 + OpenMP.
 *
 * Every query function cites the reference lines it restates (paths relative to the reference
 * tree).  The succinct primitives underneath stand in for vgteam/sdsl-lite, which is not in the
 * tree; they implement the published semantics rank(B, i) = #1 in B[0, i), select(B, j) =
 * position of the j-th 1 (j >= 1), B[i] (paper/paper.tex:133-137).
 */
#include "gcsa_oracle.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------ */
/* bit helpers                                                                                */

static inline u64 popc(u64 x) { return (u64)__builtin_popcountll(x); }
static inline u64 lo_mask(u64 bits) { return bits >= 64 ? ~(u64)0 : (((u64)1 << bits) - 1); }
static inline int get_bit(const u64* w, u64 i) { return (int)((w[i >> 6] >> (i & 63)) & 1); }

/* position (0..63) of the k-th (k >= 1) set bit of x */
static inline u64 select_in_word(u64 x, u64 k)
{
  for(u64 j = 1; j < k; j++) { x &= x - 1; }
  return (u64)__builtin_ctzll(x);
}

static u64* copy_words(const u64* src, u64 bits)
{
  u64 words = (bits + 63) / 64 + 1;
  u64* res = (u64*)calloc(words, sizeof(u64));
  if(src != NULL && bits > 0) { memcpy(res, src, ((bits + 63) / 64) * sizeof(u64)); }
  if(bits & 63) { res[bits >> 6] &= lo_mask(bits & 63); } /* clear bits past the end */
  return res;
}

/* ------------------------------------------------------------------------------------------ */
/* bit_vector_il<512> + rank_1_type (include/gcsa/gcsa.h:46): every 8 payload words are        */
/* preceded by one cumulative popcount word -> 9-word (72-byte) stride.                        */

typedef struct { u64 size; u64 blocks; u64* data; } bvil;

static void bvil_build(bvil* b, const u64* words, u64 size)
{
  b->size = size;
  b->blocks = (size >> 9) + 1;             /* rank(size) must be answerable */
  b->data = (u64*)calloc(b->blocks * 9, sizeof(u64));
  u64 total_words = (size + 63) / 64, cumul = 0;
  for(u64 blk = 0; blk < b->blocks; blk++)
  {
    u64* dst = b->data + blk * 9;
    dst[0] = cumul;
    for(u64 j = 0; j < 8; j++)
    {
      u64 w = blk * 8 + j, val = 0;
      if(w < total_words)
      {
        val = words[w];
        if(w == (size >> 6) && (size & 63)) { val &= lo_mask(size & 63); }
      }
      dst[1 + j] = val; cumul += popc(val);
    }
  }
}

static inline u64 bvil_rank(const bvil* b, u64 i)
{
  const u64* blk = b->data + (i >> 9) * 9;
  u64 res = blk[0], word = (i >> 6) & 7;
  for(u64 j = 0; j < word; j++) { res += popc(blk[1 + j]); }
  return res + popc(blk[1 + word] & lo_mask(i & 63));
}

static inline int bvil_get(const bvil* b, u64 i)
{
  return (int)((b->data[(i >> 9) * 9 + 1 + ((i >> 6) & 7)] >> (i & 63)) & 1);
}

static void bvil_free(bvil* b) { free(b->data); b->data = NULL; }

/* ------------------------------------------------------------------------------------------ */
/* plain bit_vector + rank + select_1 / select_0 (stands in for select_support_mcl).           */

#define SEL_SAMPLE 512

typedef struct
{
  u64 size, ones;
  u64* words;
  u64* cumul;       /* ones before each 512-bit block, blocks + 1 entries */
  u64 blocks;
  u64* hint1;       /* block holding the (j * SEL_SAMPLE + 1)-th one */
  u64* hint0;       /* likewise for zeros */
} selbv;

static void selbv_build(selbv* b, const u64* words, u64 size, int want_select0)
{
  b->size = size;
  b->words = copy_words(words, size);
  b->blocks = (size >> 9) + 1;
  b->cumul = (u64*)calloc(b->blocks + 1, sizeof(u64));
  u64 total_words = (size + 63) / 64;
  for(u64 blk = 0; blk < b->blocks; blk++)
  {
    u64 c = 0;
    for(u64 j = 0; j < 8; j++) { u64 w = blk * 8 + j; if(w < total_words) { c += popc(b->words[w]); } }
    b->cumul[blk + 1] = b->cumul[blk] + c;
  }
  b->ones = b->cumul[b->blocks];
  u64 h1 = b->ones / SEL_SAMPLE + 2;
  b->hint1 = (u64*)calloc(h1, sizeof(u64));
  for(u64 j = 0, blk = 0; j < h1; j++)
  {
    u64 target = j * SEL_SAMPLE + 1;  /* first block with cumul[blk + 1] >= target */
    while(blk + 1 < b->blocks && b->cumul[blk + 1] < target) { blk++; }
    b->hint1[j] = blk;
  }
  b->hint0 = NULL;
  if(want_select0)
  {
    u64 zeros = size - b->ones, h0 = zeros / SEL_SAMPLE + 2;
    b->hint0 = (u64*)calloc(h0, sizeof(u64));
    for(u64 j = 0, blk = 0; j < h0; j++)
    {
      u64 target = j * SEL_SAMPLE + 1;
      while(blk + 1 < b->blocks && ((blk + 1) << 9) - b->cumul[blk + 1] < target) { blk++; }
      b->hint0[j] = blk;
    }
  }
}

static void selbv_free(selbv* b)
{
  free(b->words); free(b->cumul); free(b->hint1); free(b->hint0);
  b->words = b->cumul = b->hint1 = b->hint0 = NULL;
}

static inline int selbv_get(const selbv* b, u64 i) { return get_bit(b->words, i); }

static inline u64 selbv_rank(const selbv* b, u64 i)
{
  u64 blk = i >> 9, res = b->cumul[blk], word = (i >> 6) & 7;
  const u64* w = b->words + blk * 8;
  for(u64 j = 0; j < word; j++) { res += popc(w[j]); }
  return res + popc(w[word] & lo_mask(i & 63));
}

/* position of the r-th one, r in 1..ones */
static inline u64 selbv_select1(const selbv* b, u64 r)
{
  u64 lo = b->hint1[(r - 1) / SEL_SAMPLE], hi = b->hint1[(r - 1) / SEL_SAMPLE + 1];
  while(lo < hi)                      /* first block with cumul[blk + 1] >= r */
  {
    u64 mid = lo + (hi - lo) / 2;
    if(b->cumul[mid + 1] >= r) { hi = mid; } else { lo = mid + 1; }
  }
  u64 rem = r - b->cumul[lo];
  const u64* w = b->words + lo * 8;
  for(u64 j = 0; ; j++)
  {
    u64 c = popc(w[j]);
    if(c >= rem) { return (lo << 9) + (j << 6) + select_in_word(w[j], rem); }
    rem -= c;
  }
}

/* position of the r-th zero, r >= 1 */
static inline u64 selbv_select0(const selbv* b, u64 r)
{
  u64 lo = b->hint0[(r - 1) / SEL_SAMPLE], hi = b->hint0[(r - 1) / SEL_SAMPLE + 1];
  while(lo < hi)
  {
    u64 mid = lo + (hi - lo) / 2;
    if(((mid + 1) << 9) - b->cumul[mid + 1] >= r) { hi = mid; } else { lo = mid + 1; }
  }
  u64 rem = r - ((lo << 9) - b->cumul[lo]);
  const u64* w = b->words + lo * 8;
  for(u64 j = 0; ; j++)
  {
    u64 inv = ~w[j], c = popc(inv);
    if(c >= rem) { return (lo << 9) + (j << 6) + select_in_word(inv, rem); }
    rem -= c;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* int_vector<0>: element i occupies bits [i * w, (i + 1) * w), LSB first.                     */

typedef struct { u64 size, width; u64* words; } packed;

static void packed_copy(packed* p, const u64* words, u64 size, u64 width)
{
  p->size = size; p->width = width;
  p->words = copy_words(words, size * width);
}

static inline u64 packed_get(const packed* p, u64 i)
{
  u64 pos = i * p->width, word = pos >> 6, shift = pos & 63;
  u64 val = p->words[word] >> shift;
  if(shift + p->width > 64) { val |= p->words[word + 1] << (64 - shift); }
  return val & lo_mask(p->width);
}

static inline void packed_set(packed* p, u64 i, u64 v)
{
  u64 pos = i * p->width, word = pos >> 6, shift = pos & 63;
  p->words[word] |= v << shift;
  if(shift + p->width > 64) { p->words[word + 1] |= v >> (64 - shift); }
}

/* ------------------------------------------------------------------------------------------ */
/* sd_vector<> (Elias-Fano, Okanohara & Sadakane): the m ones of an n-bit vector are split     */
/* into wl low bits (packed) and the high part in negated-unary (`high`, m ones and n>>wl      */
/* zeros).  rank via select_0 on high + scan of the bucket; select via select_1 on high.       */

typedef struct { u64 size, ones, wl; packed low; selbv high; } sdv;

static inline u64 hi_bit(u64 x) { return x == 0 ? 0 : 63 - (u64)__builtin_clzll(x); }

static void sdv_build(sdv* s, const u64* words, u64 size)
{
  s->size = size;
  u64 m = 0, total_words = (size + 63) / 64;
  for(u64 w = 0; w < total_words; w++)
  {
    u64 val = words[w];
    if(w == (size >> 6) && (size & 63)) { val &= lo_mask(size & 63); }
    m += popc(val);
  }
  s->ones = m;
  u64 logm = hi_bit(m) + 1, logn = hi_bit(size) + 1;
  if(logm == logn) { logm--; }
  s->wl = (m == 0 ? 0 : logn - logm);
  s->low.size = m; s->low.width = (s->wl == 0 ? 1 : s->wl);
  s->low.words = (u64*)calloc((m * s->low.width + 63) / 64 + 2, sizeof(u64));
  u64 high_len = m + (size >> s->wl) + 1;
  u64* high = (u64*)calloc((high_len + 63) / 64 + 1, sizeof(u64));
  u64 k = 0;
  for(u64 w = 0; w < total_words; w++)
  {
    u64 val = words[w];
    if(w == (size >> 6) && (size & 63)) { val &= lo_mask(size & 63); }
    while(val)
    {
      u64 pos = (w << 6) + (u64)__builtin_ctzll(val); val &= val - 1;
      if(s->wl > 0) { packed_set(&(s->low), k, pos & lo_mask(s->wl)); }
      u64 hp = (pos >> s->wl) + k;
      high[hp >> 6] |= (u64)1 << (hp & 63);
      k++;
    }
  }
  selbv_build(&(s->high), high, high_len, 1);
  free(high);
}

static void sdv_free(sdv* s) { free(s->low.words); s->low.words = NULL; selbv_free(&(s->high)); }

static inline u64 sdv_low(const sdv* s, u64 k) { return s->wl == 0 ? 0 : packed_get(&(s->low), k); }

/* number of ones in [0, i), 0 <= i <= size */
static inline u64 sdv_rank(const sdv* s, u64 i)
{
  if(s->ones == 0) { return 0; }
  if(i >= s->size) { return s->ones; }
  u64 h = i >> s->wl, l = i & lo_mask(s->wl);
  /* ones of buckets < h: position of the h-th zero minus (h - 1) zeros before it */
  u64 begin = (h == 0 ? 0 : selbv_select0(&(s->high), h) + 1 - h);
  u64 end = selbv_select0(&(s->high), h + 1) - h;   /* ones of buckets <= h */
  while(begin < end && sdv_low(s, begin) < l) { begin++; }
  return begin;
}

/* position of the r-th one, r in 1..ones */
static inline u64 sdv_select(const sdv* s, u64 r)
{
  u64 hp = selbv_select1(&(s->high), r);
  return ((hp - (r - 1)) << s->wl) | sdv_low(s, r - 1);
}

static inline int sdv_get(const sdv* s, u64 i) { return (int)(sdv_rank(s, i + 1) - sdv_rank(s, i)); }

/* ------------------------------------------------------------------------------------------ */

struct oracle_index
{
  u64 n, e, order, sigma, fast_chars;
  uint8_t char2comp[256];
  u64 C[GCSA2_MAX_SIGMA + 1];

  bvil fast_bwt[GCSA2_MAX_SIGMA];    /* comps 1..fast_chars     (gcsa.h:217-219) */
  sdv  sparse_bwt[GCSA2_MAX_SIGMA];  /* comp 0 and > fast_chars (gcsa.h:221-223, gcsa.cpp:678-686) */
  bvil edges;                        /* gcsa.h:226-227 */
  int has_samples;
  bvil sampled_paths;                /* gcsa.h:230-231 */
  packed stored_samples;             /* gcsa.h:234 */
  selbv samples;                     /* gcsa.h:235-236 */
  int has_counters;
  sdv extra_filter, extra_values;    /* SadaSparse, support.h:318-324 */
  selbv redundant;                   /* SadaCount, support.h:252-253 */

  int has_lcp;
  u64 lcp_size, lcp_branching, lcp_levels, lcp_values;
  u64* lcp_offsets;
  uint8_t* lcp_data;
};

static inline int is_fast(const oracle_index* ix, u64 comp) { return comp > 0 && comp <= ix->fast_chars; }

oracle_index* oracle_create(const gcsa2_host_view* v)
{
  if(v == NULL || v->sigma == 0 || v->sigma > GCSA2_MAX_SIGMA) { return NULL; }
  oracle_index* ix = (oracle_index*)calloc(1, sizeof(oracle_index));
  ix->n = v->path_nodes; ix->e = v->edges; ix->order = v->order;
  ix->sigma = v->sigma; ix->fast_chars = v->fast_chars;
  memcpy(ix->char2comp, v->char2comp, 256);
  for(u64 c = 0; c <= v->sigma; c++) { ix->C[c] = v->C[c]; }
  for(u64 c = 0; c < v->sigma; c++)
  {
    if(is_fast(ix, c)) { bvil_build(&(ix->fast_bwt[c]), v->bwt[c], ix->n); }
    else { sdv_build(&(ix->sparse_bwt[c]), v->bwt[c], ix->n); }
  }
  bvil_build(&(ix->edges), v->edge_bits, ix->e);
  if(v->sampled_path_bits != NULL)
  {
    ix->has_samples = 1;
    bvil_build(&(ix->sampled_paths), v->sampled_path_bits, ix->n);
    packed_copy(&(ix->stored_samples), v->stored_samples, v->sample_count, v->sample_width);
    selbv_build(&(ix->samples), v->sample_bits, v->sample_count, 0);
  }
  if(v->extra_filter_bits != NULL)
  {
    ix->has_counters = 1;
    sdv_build(&(ix->extra_filter), v->extra_filter_bits, ix->n);
    sdv_build(&(ix->extra_values), v->extra_values_bits, v->extra_values_len);
    selbv_build(&(ix->redundant), v->redundant_bits, v->redundant_len, 0);
  }
  if(v->lcp_data != NULL)
  {
    ix->has_lcp = 1;
    ix->lcp_size = v->lcp_size; ix->lcp_branching = v->lcp_branching; ix->lcp_levels = v->lcp_levels;
    ix->lcp_offsets = (u64*)malloc((v->lcp_levels + 1) * sizeof(u64));
    memcpy(ix->lcp_offsets, v->lcp_offsets, (v->lcp_levels + 1) * sizeof(u64));
    ix->lcp_values = ix->lcp_offsets[ix->lcp_levels];
    ix->lcp_data = (uint8_t*)malloc(ix->lcp_values + 1);
    memcpy(ix->lcp_data, v->lcp_data, ix->lcp_values);
  }
  return ix;
}

void oracle_destroy(oracle_index* ix)
{
  if(ix == NULL) { return; }
  for(u64 c = 0; c < ix->sigma; c++)
  {
    if(is_fast(ix, c)) { bvil_free(&(ix->fast_bwt[c])); } else { sdv_free(&(ix->sparse_bwt[c])); }
  }
  bvil_free(&(ix->edges));
  if(ix->has_samples) { bvil_free(&(ix->sampled_paths)); free(ix->stored_samples.words); selbv_free(&(ix->samples)); }
  if(ix->has_counters) { sdv_free(&(ix->extra_filter)); sdv_free(&(ix->extra_values)); selbv_free(&(ix->redundant)); }
  free(ix->lcp_offsets); free(ix->lcp_data);
  free(ix);
}

void oracle_free(void* p) { free(p); }
int oracle_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------------------------ */
/* Range (include/gcsa/utils.h:84-117)                                                         */

static inline int range_empty(u64 sp, u64 ep) { return sp + 1 > ep + 1; }   /* utils.h:93-96 */
static inline u64 range_length(u64 sp, u64 ep) { return ep + 1 - sp; }      /* utils.h:88-91 */

/* rank[comp](i): fast_rank for comps 1..fast_chars, sparse_rank otherwise (gcsa.h:157-158) */
static inline u64 bwt_rank(const oracle_index* ix, u64 comp, u64 i)
{
  return is_fast(ix, comp) ? bvil_rank(&(ix->fast_bwt[comp]), i) : sdv_rank(&(ix->sparse_bwt[comp]), i);
}

static inline int bwt_get(const oracle_index* ix, u64 comp, u64 i)
{
  return is_fast(ix, comp) ? bvil_get(&(ix->fast_bwt[comp]), i) : sdv_get(&(ix->sparse_bwt[comp]), i);
}

/* private LF(rank, i, comp) = C[comp] + rank[comp](i)  (gcsa.h:262-266) */
static inline u64 lf_edge(const oracle_index* ix, u64 i, u64 comp) { return ix->C[comp] + bwt_rank(ix, comp, i); }

/* pathNodeRange (gcsa.h:253-258) */
static inline void path_node_range(const oracle_index* ix, u64* sp, u64* ep)
{
  *sp = bvil_rank(&(ix->edges), *sp);
  *ep = bvil_rank(&(ix->edges), *ep);
}

/* GCSA::charRange (gcsa.h:150-153) over gcsa::charRange(alpha, comp) (utils.h:414-419):
 * (C[c], C[c+1] - 1) converted without an emptiness check. */
void oracle_char_range(const oracle_index* ix, uint8_t comp, u64* sp, u64* ep)
{
  *sp = ix->C[comp]; *ep = ix->C[comp + 1] - 1;
  path_node_range(ix, sp, ep);
}

/* GCSA::LF(range, comp) (gcsa.h:155-162): an empty result is returned in EDGE space. */
void oracle_lf_range(const oracle_index* ix, u64* sp, u64* ep, uint8_t comp)
{
  u64 a = lf_edge(ix, *sp, comp);            /* gcsa.h:271 */
  u64 b = lf_edge(ix, *ep + 1, comp) - 1;    /* gcsa.h:272 */
  *sp = a; *ep = b;
  if(range_empty(a, b)) { return; }          /* gcsa.h:160 */
  path_node_range(ix, sp, ep);               /* gcsa.h:161 */
}

/* GCSA::find(begin, end) (gcsa.h:96-110) */
void oracle_find(const oracle_index* ix, const uint8_t* pattern, u64 length, u64* sp, u64* ep)
{
  if(length == 0 || ix->n == 0) { *sp = 0; *ep = ix->n - 1; return; }     /* gcsa.h:99 */
  u64 i = length - 1;
  oracle_char_range(ix, ix->char2comp[pattern[i]], sp, ep);                /* gcsa.h:101-102 */
  while(!range_empty(*sp, *ep) && i > 0)                                   /* gcsa.h:103 */
  {
    i--;
    oracle_lf_range(ix, sp, ep, ix->char2comp[pattern[i]]);                /* gcsa.h:105-106 */
  }
}

/* GCSA::LF(path_node) (gcsa.h:165-183): first incoming edge, fast comps first. */
uint64_t oracle_lf_node(const oracle_index* ix, u64 node)
{
  for(u64 comp = 1; comp <= ix->fast_chars; comp++)
  {
    if(bvil_get(&(ix->fast_bwt[comp]), node)) { return bvil_rank(&(ix->edges), lf_edge(ix, node, comp)); }
  }
  for(u64 comp = ix->fast_chars + 1; comp < ix->sigma; comp++)
  {
    if(sdv_get(&(ix->sparse_bwt[comp]), node)) { return bvil_rank(&(ix->edges), lf_edge(ix, node, comp)); }
  }
  return bvil_rank(&(ix->edges), lf_edge(ix, node, 0));
}

/* GCSA::LF_fast (src/gcsa.cpp:742-766), GCSA::LF_all (src/gcsa.cpp:768-798) */
void oracle_lf_all(const oracle_index* ix, u64 sp, u64 ep, int all, u64* out)
{
  for(u64 c = 0; c < ix->sigma; c++) { out[2 * c] = 1; out[2 * c + 1] = 0; }
  if(range_empty(sp, ep)) { return; }
  u64 limit = (all ? ix->sigma - 2 : ix->fast_chars);   /* comps 1..limit */
  if(sp == ep)  /* single path node: bit probes (gcsa.cpp:748-757, 774-791) */
  {
    for(u64 c = 1; c <= limit; c++)
    {
      if(bwt_get(ix, c, sp)) { out[2 * c] = out[2 * c + 1] = bvil_rank(&(ix->edges), lf_edge(ix, sp, c)); }
    }
  }
  else          /* general case (gcsa.cpp:758-765, 792-797) */
  {
    for(u64 c = 1; c <= limit; c++)
    {
      u64 a = sp, b = ep;
      oracle_lf_range(ix, &a, &b, (uint8_t)c);
      out[2 * c] = a; out[2 * c + 1] = b;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* counting (src/gcsa.cpp:802-809)                                                             */

/* SadaSparse::count (support.h:329-335) */
static inline u64 sada_sparse_count(const oracle_index* ix, u64 sp, u64 ep)
{
  u64 a = sdv_rank(&(ix->extra_filter), sp);
  u64 b = sdv_rank(&(ix->extra_filter), ep + 1);
  if(b <= a) { return 0; }
  return (sdv_select(&(ix->extra_values), b) + 1) - (a > 0 ? sdv_select(&(ix->extra_values), a) + 1 : 0);
}

/* SadaCount::count (support.h:255-258) */
static inline u64 sada_count(const oracle_index* ix, u64 sp, u64 ep)
{
  return (selbv_select1(&(ix->redundant), ep + 1) - ep)
       - (sp > 0 ? selbv_select1(&(ix->redundant), sp) + 1 - sp : 0);
}

uint64_t oracle_count(const oracle_index* ix, u64 sp, u64 ep)
{
  if(range_empty(sp, ep) || ep >= ix->n) { return 0; }                  /* gcsa.cpp:805 */
  u64 res = sada_sparse_count(ix, sp, ep) + range_length(sp, ep);       /* gcsa.cpp:806 */
  if(ep > sp) { res -= sada_count(ix, sp, ep - 1); }                    /* gcsa.cpp:807 */
  return res;
}

/* ------------------------------------------------------------------------------------------ */
/* locate (src/gcsa.cpp:813-896)                                                               */

typedef struct { u64* data; u64 size, cap; } u64vec;

static inline void vec_push(u64vec* v, u64 x)
{
  if(v->size == v->cap)
  {
    v->cap = (v->cap == 0 ? 16 : 2 * v->cap);
    v->data = (u64*)realloc(v->data, v->cap * sizeof(u64));
  }
  v->data[v->size++] = x;
}

static int cmp_u64(const void* a, const void* b)
{
  u64 x = *(const u64*)a, y = *(const u64*)b;
  return (x > y) - (x < y);
}

/* removeDuplicates (utils.h:350-357) */
static void vec_sort_unique(u64vec* v)
{
  if(v->size == 0) { return; }
  qsort(v->data, v->size, sizeof(u64), cmp_u64);
  u64 tail = 1;
  for(u64 i = 1; i < v->size; i++) { if(v->data[i] != v->data[tail - 1]) { v->data[tail++] = v->data[i]; } }
  v->size = tail;
}

int oracle_sampled(const oracle_index* ix, u64 node) { return bvil_get(&(ix->sampled_paths), node); }   /* gcsa.h:191 */

uint64_t oracle_first_sample(const oracle_index* ix, u64 node)                                         /* gcsa.h:202-206 */
{
  u64 r = bvil_rank(&(ix->sampled_paths), node);
  return (r > 0 ? selbv_select1(&(ix->samples), r) + 1 : 0);
}

int oracle_last_sample(const oracle_index* ix, u64 i) { return selbv_get(&(ix->samples), i); }         /* gcsa.h:208 */
uint64_t oracle_sample(const oracle_index* ix, u64 i) { return packed_get(&(ix->stored_samples), i); } /* gcsa.h:210 */

/* GCSA::locateInternal (src/gcsa.cpp:880-896) */
static void locate_internal(const oracle_index* ix, u64 node, u64vec* out)
{
  u64 steps = 0;
  while(!oracle_sampled(ix, node)) { node = oracle_lf_node(ix, node); steps++; }
  u64 s = oracle_first_sample(ix, node);
  do { vec_push(out, oracle_sample(ix, s) + steps); s++; }
  while(!oracle_last_sample(ix, s - 1));
}

/* GCSA::locate(range, results, append = false, sort) (src/gcsa.cpp:827-842) */
static void locate_range(const oracle_index* ix, u64 sp, u64 ep, int sort, u64vec* out)
{
  out->size = 0;
  if(range_empty(sp, ep) || ep >= ix->n) { return; }
  for(u64 i = sp; i <= ep; i++) { locate_internal(ix, i, out); }
  if(sort) { vec_sort_unique(out); }
}

uint64_t* oracle_locate(const oracle_index* ix, u64 sp, u64 ep, int sort, u64* count)
{
  u64vec v = { NULL, 0, 0 };
  locate_range(ix, sp, ep, sort, &v);
  *count = v.size;
  if(v.data == NULL) { v.data = (u64*)malloc(sizeof(u64)); }
  return v.data;
}

/* std::mt19937_64 (ISO C++ [rand.predef]: w=64 n=312 m=156 r=31 a=0xB5026F5AA96619E9 u=29
 * d=0x5555555555555555 s=17 b=0x71D67FFFEDA60000 t=37 c=0xFFF7EEE000000000 l=43 f=6364136223846793005) */
typedef struct { u64 mt[312]; int idx; } mt64;

static void mt64_seed(mt64* g, u64 seed)
{
  g->mt[0] = seed;
  for(int i = 1; i < 312; i++) { g->mt[i] = 6364136223846793005ULL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (u64)i; }
  g->idx = 312;
}

static u64 mt64_next(mt64* g)
{
  if(g->idx >= 312)
  {
    for(int i = 0; i < 312; i++)
    {
      u64 x = (g->mt[i] & 0xFFFFFFFF80000000ULL) | (g->mt[(i + 1) % 312] & 0x7FFFFFFFULL);
      u64 xa = x >> 1;
      if(x & 1) { xa ^= 0xB5026F5AA96619E9ULL; }
      g->mt[i] = g->mt[(i + 156) % 312] ^ xa;
    }
    g->idx = 0;
  }
  u64 y = g->mt[g->idx++];
  y ^= (y >> 29) & 0x5555555555555555ULL;
  y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
  y ^= (y << 37) & 0xFFF7EEE000000000ULL;
  y ^= (y >> 43);
  return y;
}

/* GCSA::locate(range, max_positions, results) (src/gcsa.cpp:844-878).  The unordered_set of the
 * reference is replaced by a sorted unique vector: its iteration order never reaches the output
 * because deterministicShuffle sorts first (utils.h:359-370) and the result is sorted last. */
uint64_t* oracle_locate_max(const oracle_index* ix, u64 sp, u64 ep, u64 max_positions, u64* count)
{
  u64vec res = { NULL, 0, 0 };
  u64 total = oracle_count(ix, sp, ep);
  if(total == 0) { *count = 0; return (u64*)malloc(sizeof(u64)); }
  if(max_positions > total) { max_positions = total; }
  mt64 rng; mt64_seed(&rng, sp ^ ep);
  if(max_positions >= total / 2) { locate_range(ix, sp, ep, 1, &res); }
  else
  {
    u64vec found = { NULL, 0, 0 }, tmp = { NULL, 0, 0 };
    while(found.size < max_positions)
    {
      u64 pos = sp + mt64_next(&rng) % range_length(sp, ep);
      tmp.size = 0;
      locate_internal(ix, pos, &tmp);
      for(u64 i = 0; i < tmp.size; i++) { vec_push(&found, tmp.data[i]); }
      vec_sort_unique(&found);
    }
    res = found; free(tmp.data);
  }
  if(res.size > max_positions)
  {
    qsort(res.data, res.size, sizeof(u64), cmp_u64);           /* deterministicShuffle sorts first */
    for(u64 i = res.size; i > 0; i--)
    {
      u64 j = mt64_next(&rng) % i, t = res.data[i - 1];
      res.data[i - 1] = res.data[j]; res.data[j] = t;
    }
    res.size = max_positions;
  }
  qsort(res.data, res.size, sizeof(u64), cmp_u64);
  *count = res.size;
  if(res.data == NULL) { res.data = (u64*)malloc(sizeof(u64)); }
  return res.data;
}

/* ------------------------------------------------------------------------------------------ */
/* LCPArray (include/gcsa/lcp.h, src/lcp.cpp)                                                  */

static inline u64 lcp_at(const oracle_index* ix, u64 i) { return ix->lcp_data[i]; }

/* range-minimum-tree helpers (src/lcp.cpp:152-200) */
static inline u64 rmt_root(const oracle_index* ix) { return ix->lcp_values - 1; }
static inline u64 rmt_parent(const oracle_index* ix, u64 node, u64 level)
{ return ix->lcp_offsets[level + 1] + (node - ix->lcp_offsets[level]) / ix->lcp_branching; }
static inline u64 rmt_first_sibling(const oracle_index* ix, u64 node, u64 level)
{ return node - (node - ix->lcp_offsets[level]) % ix->lcp_branching; }
static inline u64 rmt_last_sibling(const oracle_index* ix, u64 first_child, u64 level)
{
  u64 a = ix->lcp_offsets[level + 1], b = first_child + ix->lcp_branching;
  return (a < b ? a : b) - 1;
}
static inline u64 rmt_first_child(const oracle_index* ix, u64 node, u64 level)
{ return ix->lcp_offsets[level - 1] + (node - ix->lcp_offsets[level]) * ix->lcp_branching; }
static inline u64 rmt_last_child(const oracle_index* ix, u64 node, u64 level)
{ return rmt_last_sibling(ix, rmt_first_child(ix, node, level), level - 1); }
static inline u64 rmt_level(const oracle_index* ix, u64 node)
{ u64 level = 0; while(ix->lcp_offsets[level + 1] <= node) { level++; } return level; }

static inline int sv_cmp(int equal, u64 a, u64 b) { return equal ? (a <= b) : (a < b); }

/* last value satisfying cmp in [from, to) (src/lcp.cpp:333-343) */
static int psv_scan(const oracle_index* ix, u64 from, u64 to, u64 val, int equal, u64* pos, u64* v)
{
  while(to > from)
  {
    to--;
    if(sv_cmp(equal, lcp_at(ix, to), val)) { *pos = to; *v = lcp_at(ix, to); return 1; }
  }
  return 0;
}

/* first value satisfying cmp in [from, to] (src/lcp.cpp:390-399) */
static int nsv_scan(const oracle_index* ix, u64 from, u64 to, u64 val, int equal, u64* pos, u64* v)
{
  for(u64 i = from; i <= to; i++)
  {
    if(sv_cmp(equal, lcp_at(ix, i), val)) { *pos = i; *v = lcp_at(ix, i); return 1; }
  }
  return 0;
}

/* psv / psev (src/lcp.cpp:345-382) */
static void lcp_psv(const oracle_index* ix, u64 to, int equal, u64* rpos, u64* rval)
{
  *rpos = *rval = ix->lcp_values;   /* notFound() (lcp.h:178) */
  if(to == 0 || to >= ix->lcp_size) { return; }
  u64 level = 0, val = lcp_at(ix, to);
  int found = 0;
  while(to != rmt_root(ix))
  {
    found = psv_scan(ix, rmt_first_sibling(ix, to, level), to, val, equal, rpos, rval);
    if(found) { break; }
    to = rmt_parent(ix, to, level); level++;
  }
  if(!found) { *rpos = *rval = ix->lcp_values; return; }
  while(level > 0)
  {
    u64 from = rmt_first_child(ix, *rpos, level); level--;
    psv_scan(ix, from, rmt_last_sibling(ix, from, level) + 1, val, equal, rpos, rval);
  }
}

/* nsv / nsev (src/lcp.cpp:401-438) */
static void lcp_nsv(const oracle_index* ix, u64 from, int equal, u64* rpos, u64* rval)
{
  *rpos = *rval = ix->lcp_values;
  if(from + 1 >= ix->lcp_size) { return; }
  u64 level = 0, val = lcp_at(ix, from);
  int found = 0;
  while(from != rmt_root(ix))
  {
    found = nsv_scan(ix, from + 1, rmt_last_sibling(ix, from, level), val, equal, rpos, rval);
    if(found) { break; }
    from = rmt_parent(ix, from, level); level++;
  }
  if(!found) { *rpos = *rval = ix->lcp_values; return; }
  while(level > 0)
  {
    from = rmt_first_child(ix, *rpos, level); level--;
    nsv_scan(ix, from, rmt_last_sibling(ix, from, level), val, equal, rpos, rval);
  }
}

void oracle_sv(const oracle_index* ix, int op, u64 pos, u64* res_pos, u64* res_val)
{
  if(op < 2) { lcp_psv(ix, pos, op & 1, res_pos, res_val); }
  else { lcp_nsv(ix, pos, op & 1, res_pos, res_val); }
}

static inline void update_res(const oracle_index* ix, u64* rpos, u64* rval, u64 i)   /* lcp.cpp:442-446 */
{
  if(lcp_at(ix, i) < *rval) { *rpos = i; *rval = lcp_at(ix, i); }
}

/* LCPArray::rmq (src/lcp.cpp:448-513): leftmost minimum of LCP[sp..ep]. */
void oracle_rmq(const oracle_index* ix, u64 sp, u64 ep, u64* rpos, u64* rval)
{
  if(sp > ep || ep >= ix->lcp_size) { *rpos = *rval = ix->lcp_values; return; }
  if(sp == ep) { *rpos = sp; *rval = lcp_at(ix, sp); return; }

  *rpos = ix->lcp_values; *rval = ix->lcp_size;
  u64 level = 0, left = sp, right = ep;
  /* tail stack: at most `branching` entries per level */
  u64 tail_cap = (ix->lcp_levels + 1) * ix->lcp_branching, tail_size = 0;
  u64* tail = (u64*)malloc(2 * tail_cap * sizeof(u64));
  while(1)
  {
    u64 left_par = rmt_parent(ix, left, level), right_par = rmt_parent(ix, right, level);
    if(left_par == right_par)
    {
      for(u64 i = left; i <= right; i++) { update_res(ix, rpos, rval, i); }
      break;
    }
    u64 left_child = rmt_first_child(ix, left_par, level + 1);
    if(left != left_child)
    {
      u64 last_child = rmt_last_sibling(ix, left_child, level);
      for(u64 i = left; i <= last_child; i++) { update_res(ix, rpos, rval, i); }
      left_par++;
    }
    u64 right_child = rmt_last_child(ix, right_par, level + 1);
    if(right != right_child)
    {
      u64 first_child = rmt_first_sibling(ix, right_child, level);
      for(u64 i = right; ; i--)
      {
        tail[2 * tail_size] = i; tail[2 * tail_size + 1] = lcp_at(ix, i); tail_size++;
        if(i == first_child) { break; }
      }
      right_par--;
    }
    if(left_par >= right_par)
    {
      if(left_par == right_par) { update_res(ix, rpos, rval, left_par); }
      break;
    }
    left = left_par; right = right_par; level++;
  }
  while(tail_size > 0)
  {
    tail_size--;
    if(tail[2 * tail_size + 1] < *rval) { *rpos = tail[2 * tail_size]; *rval = tail[2 * tail_size + 1]; }
  }
  free(tail);
  level = rmt_level(ix, *rpos);
  while(level > 0)
  {
    *rpos = rmt_first_child(ix, *rpos, level); level--;
    while(lcp_at(ix, *rpos) != *rval) { (*rpos)++; }
  }
}

/* LCPArray::nodeFor (lcp.h:163-175) + LCPArray::parent (src/lcp.cpp:276-301) */
void oracle_parent(const oracle_index* ix, u64 sp, u64 ep, gcsa2_stnode* out)
{
  u64 left_lcp = lcp_at(ix, sp);
  u64 right_lcp = (ep + 1 < ix->lcp_size ? lcp_at(ix, ep + 1) : 0);
  if(sp == 0 && ep == ix->lcp_size - 1)        /* node == root() (lcp.cpp:278, lcp.h:137) */
  {
    out->sp = 0; out->ep = ix->lcp_size - 1; out->left_lcp = 0; out->right_lcp = 0; out->node_lcp = 0;
    return;
  }
  u64 node_lcp = (left_lcp > right_lcp ? left_lcp : right_lcp);
  u64 lpos = sp, lval = left_lcp, rpos = ep + 1, rval = right_lcp;
  if(left_lcp == node_lcp)
  {
    lcp_psv(ix, sp, 0, &lpos, &lval);
    if(lpos == ix->lcp_values && lval == ix->lcp_values) { lpos = 0; lval = 0; }
  }
  if(right_lcp == node_lcp)
  {
    lcp_nsv(ix, ep + 1, 0, &rpos, &rval);
    if(rpos == ix->lcp_values && rval == ix->lcp_values) { rpos = ix->lcp_size; rval = 0; }
  }
  out->sp = lpos; out->ep = rpos - 1; out->left_lcp = lval; out->right_lcp = rval; out->node_lcp = node_lcp;
}

/* LCPArray::depth(range) (src/lcp.cpp:319-325) */
uint64_t oracle_depth(const oracle_index* ix, u64 sp, u64 ep)
{
  if(range_length(sp, ep) <= 1) { return GCSA2_UNKNOWN; }
  u64 pos, val;
  oracle_rmq(ix, sp + 1, ep, &pos, &val);
  return (pos == ix->lcp_values && val == ix->lcp_values ? GCSA2_UNKNOWN : val);
}

/* ------------------------------------------------------------------------------------------ */
/* batched drivers                                                                             */

double oracle_find_batch(const oracle_index* ix, const uint8_t* patterns, const u64* offsets,
                         u64 nq, u64* ranges, int threads)
{
  double start = omp_get_wtime();
  if(threads <= 1)
  {
    for(u64 q = 0; q < nq; q++)
    {
      oracle_find(ix, patterns + offsets[q], offsets[q + 1] - offsets[q], ranges + 2 * q, ranges + 2 * q + 1);
    }
  }
  else
  {
    #pragma omp parallel for schedule(static) num_threads(threads)
    for(u64 q = 0; q < nq; q++)
    {
      oracle_find(ix, patterns + offsets[q], offsets[q + 1] - offsets[q], ranges + 2 * q, ranges + 2 * q + 1);
    }
  }
  return omp_get_wtime() - start;
}

double oracle_lf_batch(const oracle_index* ix, const u64* in, const uint8_t* comps, u64 nq, u64* out, int threads)
{
  double start = omp_get_wtime();
  #pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++)
  {
    u64 sp = in[2 * q], ep = in[2 * q + 1];
    oracle_lf_range(ix, &sp, &ep, comps[q]);
    out[2 * q] = sp; out[2 * q + 1] = ep;
  }
  return omp_get_wtime() - start;
}

double oracle_count_batch(const oracle_index* ix, const u64* ranges, u64 nq, u64* counts, int threads)
{
  double start = omp_get_wtime();
  #pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++) { counts[q] = oracle_count(ix, ranges[2 * q], ranges[2 * q + 1]); }
  return omp_get_wtime() - start;
}

double oracle_parent_batch(const oracle_index* ix, const u64* ranges, u64 nq, gcsa2_stnode* out, int threads)
{
  double start = omp_get_wtime();
  #pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++) { oracle_parent(ix, ranges[2 * q], ranges[2 * q + 1], out + q); }
  return omp_get_wtime() - start;
}

double oracle_depth_batch(const oracle_index* ix, const u64* ranges, u64 nq, u64* out, int threads)
{
  double start = omp_get_wtime();
  #pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++) { out[q] = oracle_depth(ix, ranges[2 * q], ranges[2 * q + 1]); }
  return omp_get_wtime() - start;
}

double oracle_locate_batch(const oracle_index* ix, const u64* ranges, u64 nq, u64* offsets, u64** values, int threads)
{
  double start = omp_get_wtime();
  u64** parts = (u64**)calloc(nq > 0 ? nq : 1, sizeof(u64*));
  u64* sizes = (u64*)calloc(nq > 0 ? nq : 1, sizeof(u64));
  #pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++) { parts[q] = oracle_locate(ix, ranges[2 * q], ranges[2 * q + 1], 1, sizes + q); }
  offsets[0] = 0;
  for(u64 q = 0; q < nq; q++) { offsets[q + 1] = offsets[q] + sizes[q]; }
  u64* all = (u64*)malloc((offsets[nq] > 0 ? offsets[nq] : 1) * sizeof(u64));
  for(u64 q = 0; q < nq; q++) { memcpy(all + offsets[q], parts[q], sizes[q] * sizeof(u64)); free(parts[q]); }
  free(parts); free(sizes);
  *values = all;
  return omp_get_wtime() - start;
}

/* Distinct device rank blocks touched by find(): same control flow as oracle_find, but instead of
 * bytes moved on the CPU it counts the blocks of a `block_bits`-per-block device layout. */
void oracle_find_traffic(const oracle_index* ix, const uint8_t* patterns, const u64* offsets, u64 nq,
                         u64 block_bits, u64* blocks_touched, u64* lf_steps)
{
  u64 blocks = 0, steps = 0;
  #pragma omp parallel for schedule(static) reduction(+:blocks, steps)
  for(u64 q = 0; q < nq; q++)
  {
    const uint8_t* p = patterns + offsets[q];
    u64 len = offsets[q + 1] - offsets[q];
    if(len == 0 || ix->n == 0) { continue; }
    u64 i = len - 1, comp = ix->char2comp[p[i]];
    u64 a = ix->C[comp], b = ix->C[comp + 1] - 1;
    blocks += 1 + (a / block_bits != b / block_bits);
    u64 sp = bvil_rank(&(ix->edges), a), ep = bvil_rank(&(ix->edges), b);
    while(!range_empty(sp, ep) && i > 0)
    {
      i--; steps++;
      comp = ix->char2comp[p[i]];
      blocks += 1 + (sp / block_bits != (ep + 1) / block_bits);
      a = lf_edge(ix, sp, comp); b = lf_edge(ix, ep + 1, comp) - 1;
      sp = a; ep = b;
      if(range_empty(a, b)) { break; }
      blocks += 1 + (a / block_bits != b / block_bits);
      sp = bvil_rank(&(ix->edges), a); ep = bvil_rank(&(ix->edges), b);
    }
  }
  *blocks_touched = blocks; *lf_steps = steps;
}

/* ------------------------------------------------------------------------------------------ */
/* countKMers (src/algorithms.cpp:364-421): number of distinct k-mers = non-empty states at     */
/* depth k of the search tree rooted at (0, n - 1), expanding with LF_fast (bases only) or       */
/* LF_all (include_Ns).  Restated as an explicit-stack DFS per seed, OpenMP over seeds.          */

typedef struct { u64 sp, ep, k; } kstate;

static u64 count_subtree(const oracle_index* ix, kstate root, u64 k, int include_ns, kstate** seeds, u64* nseeds, u64* cap)
{
  u64 limit = (include_ns ? ix->sigma : ix->fast_chars + 2), count = 0;          /* algorithms.cpp:369 */
  u64 stack_cap = 64, top = 0;
  kstate* stack = (kstate*)malloc(stack_cap * sizeof(kstate));
  u64* pred = (u64*)malloc(2 * ix->sigma * sizeof(u64));
  stack[top++] = root;
  while(top > 0)
  {
    kstate cur = stack[--top];
    if(range_empty(cur.sp, cur.ep)) { continue; }                                   /* :373 */
    if(cur.k == k)                                                                  /* report */
    {
      if(seeds != NULL)
      {
        if(*nseeds == *cap) { *cap = (*cap == 0 ? 64 : 2 * *cap); *seeds = (kstate*)realloc(*seeds, *cap * sizeof(kstate)); }
        (*seeds)[(*nseeds)++] = cur;
      }
      count++;
    }
    if(cur.k < k)                                                                   /* expand */
    {
      oracle_lf_all(ix, cur.sp, cur.ep, include_ns, pred);                          /* :377-378 */
      for(u64 comp = 1; comp + 1 < limit; comp++)
      {
        if(top == stack_cap) { stack_cap *= 2; stack = (kstate*)realloc(stack, stack_cap * sizeof(kstate)); }
        kstate next = { pred[2 * comp], pred[2 * comp + 1], cur.k + 1 };
        stack[top++] = next;
      }
    }
  }
  free(stack); free(pred);
  return count;
}

uint64_t oracle_count_kmers(const oracle_index* ix, uint64_t k, int include_ns, int force, uint64_t seed_length, int threads)
{
  if(k == 0) { return 1; }                                                          /* :390 */
  if(k > ix->order && !force) { return 0; }                                         /* :391-395 */
  kstate* seeds = NULL; u64 nseeds = 0, cap = 0;
  kstate root = { 0, ix->n - 1, 0 };
  count_subtree(ix, root, (k < seed_length ? k : seed_length), include_ns, &seeds, &nseeds, &cap);   /* :398-405 */
  u64 result = 0;
  #pragma omp parallel for schedule(dynamic, 1) reduction(+:result) num_threads(threads > 0 ? threads : 1)
  for(u64 i = 0; i < nseeds; i++) { result += count_subtree(ix, seeds[i], k, include_ns, NULL, NULL, NULL); }   /* :408-418 */
  free(seeds);
  return result;
}

/* ------------------------------------------------------------------------------------------ */
/* Matching statistics = the LF + parent interplay vg's MEM finder drives (paper/paper.tex:344  */
/* "we can search for maximal exact matches by using LF-mapping and parent queries", after      */
/* Ohlebusch et al. 2010; the same interplay is what verifyIndex checks, algorithms.cpp:146-167).*/
/* Composition of the restated reference primitives only: GCSA::LF(range, comp) (gcsa.h:155-162) */
/* and LCPArray::parent(range) (lcp.cpp:276-301).  Scanning the pattern right to left, ms[i] is  */
/* the length of the longest match starting at i that the index reports, (sp, ep) its range for  */
/* i = 0; *fallbacks counts parent() calls.                                                       */
void oracle_match_stats(const oracle_index* ix, const uint8_t* pattern, u64 len, uint16_t* ms,
                        u64* sp_out, u64* ep_out, u64* fallbacks)
{
  u64 sp = 0, ep = ix->n - 1, depth = 0, calls = 0;
  for(u64 i = len; i-- > 0; )
  {
    uint8_t comp = ix->char2comp[pattern[i]];
    while(1)
    {
      u64 a = sp, b = ep;
      oracle_lf_range(ix, &a, &b, comp);
      if(!range_empty(a, b)) { sp = a; ep = b; depth++; break; }
      if(sp == 0 && ep == ix->n - 1) { depth = 0; break; }          /* at the root: no such character */
      gcsa2_stnode node;
      oracle_parent(ix, sp, ep, &node); calls++;
      sp = node.sp; ep = node.ep; depth = node.node_lcp;
    }
    ms[i] = (uint16_t)(depth > 65535 ? 65535 : depth);
  }
  *sp_out = sp; *ep_out = ep; *fallbacks = calls;
}

double oracle_match_stats_batch(const oracle_index* ix, const uint8_t* patterns, const u64* offsets, u64 nq,
                                uint16_t* ms, u64* ranges, u64* fallbacks, int threads)
{
  double start = omp_get_wtime();
  #pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for(u64 q = 0; q < nq; q++)
  {
    oracle_match_stats(ix, patterns + offsets[q], offsets[q + 1] - offsets[q], ms + offsets[q],
                       ranges + 2 * q, ranges + 2 * q + 1, fallbacks + q);
  }
  return omp_get_wtime() - start;
}

/* ------------------------------------------------------------------------------------------ */
/* compareKMers (src/algorithms.cpp:505-616): simultaneous search of two indexes; result =       */
/* { k-mers in both, only in the left index, only in the right index }.  Counts only (the        */
/* reference can also dump the unshared k-mers to files, algorithms.cpp:606-610).                 */

/* KMerComparisonState (algorithms.cpp:425-457): the two ranges, the depth and the k-mer, 3 bits per
 * comp, character i of the backward extension at bits [3i, 3i+3) of kmer[0..2]. */
typedef struct { u64 lsp, lep, rsp, rep, k, kmer[3]; } cstate;

static void cstate_set(cstate* st, u64 i, u64 comp)                                        /* :451-457 */
{
  u64 offset = (i * 3) >> 6, bit = (i * 3) & 63;
  st->kmer[offset] |= comp << bit;
  if(bit > 61) { st->kmer[offset + 1] |= comp >> (64 - bit); }
}

/* records != NULL: *left_records / *right_records receive malloc'ed arrays of 8-u64 states (the structs
 * the reference writes to output.left / output.right, :606-610), result[1] / result[2] of them. */
void oracle_compare_kmers_records(const oracle_index* left, const oracle_index* right, uint64_t k, int include_ns, int force,
                                  uint64_t* result, uint64_t** left_records, uint64_t** right_records)
{
  result[0] = result[1] = result[2] = 0;
  if(left_records) { *left_records = NULL; }
  if(right_records) { *right_records = NULL; }
  if(k == 0) { result[0] = 1; return; }                                                   /* :539 */
  if((k > left->order || k > right->order) && !force) { return; }                         /* :540-549 */
  if(k > 64) { return; }                                                                  /* :550-554, MAX_K */
  if(left->sigma != right->sigma || left->fast_chars != right->fast_chars) { return; }    /* :556-560 */
  u64 limit = (include_ns ? left->sigma : left->fast_chars + 2);                           /* :511 */
  u64 cap = 1024, top = 0, lcap = 0, rcap = 0;
  cstate* stack = (cstate*)malloc(cap * sizeof(cstate));
  u64* lp = (u64*)malloc(2 * left->sigma * sizeof(u64));
  u64* rp = (u64*)malloc(2 * right->sigma * sizeof(u64));
  cstate root = { 0, left->n - 1, 0, right->n - 1, 0, {0, 0, 0} };
  stack[top++] = root;
  while(top > 0)
  {
    cstate cur = stack[--top];
    int le = range_empty(cur.lsp, cur.lep), re = range_empty(cur.rsp, cur.rep);
    if(le && re) { continue; }                                                             /* :515 */
    if(cur.k == k)                                                                         /* report, :473-491 */
    {
      if(!le && !re) { result[0]++; }
      else
      {
        uint64_t** dst = (!le ? left_records : right_records);
        u64* count = (!le ? &result[1] : &result[2]);
        u64* capacity = (!le ? &lcap : &rcap);
        if(dst != NULL)
        {
          if(*count == *capacity) { *capacity = (*capacity ? 2 * *capacity : 256); *dst = (u64*)realloc(*dst, *capacity * sizeof(cstate)); }
          memcpy(*dst + 8 * *count, &cur, sizeof(cstate));
        }
        (*count)++;
      }
      continue;
    }
    oracle_lf_all(left, cur.lsp, cur.lep, include_ns, lp);                                 /* :519-526 */
    oracle_lf_all(right, cur.rsp, cur.rep, include_ns, rp);
    for(u64 comp = 1; comp + 1 < limit; comp++)
    {
      if(top == cap) { cap *= 2; stack = (cstate*)realloc(stack, cap * sizeof(cstate)); }
      cstate next = { lp[2 * comp], lp[2 * comp + 1], rp[2 * comp], rp[2 * comp + 1], cur.k + 1, { cur.kmer[0], cur.kmer[1], cur.kmer[2] } };
      cstate_set(&next, cur.k, comp);
      stack[top++] = next;
    }
  }
  free(stack); free(lp); free(rp);
}

void oracle_compare_kmers(const oracle_index* left, const oracle_index* right, uint64_t k, int include_ns, int force,
                          uint64_t* result)
{
  oracle_compare_kmers_records(left, right, k, include_ns, force, result, NULL, NULL);
}
