// builder.cpp -- scalable construction of the order-K maximally pruned de Bruijn graph of an
// input graph, as a lexicographically sorted path-node table (workload / test infrastructure;
// NOT part of the shipped query engine, and not the reference's disk-based prefix-doubling
// algorithm of src/gcsa.cpp:447-724, src/path_graph.cpp).
//
// Method: level-synchronous trie refinement, straight from the paper's definitions
// (paper/paper.tex:246-254 path graph, :288-299 pruning lemma and maximal pruning):
//   1. A trie node at depth d is a distinct path label p of length d; its members are the pairs
//      (start position v, last position u) of paths spelling p.  A node whose members all share
//      one start is a leaf (every order-K extension has the value set {v}); a node at depth K is
//      a leaf; every other node is expanded by one character.
//   2. Bottom-up, a node is "uniform" if all leaves below it carry the same value set.  The
//      final path nodes (= keys of the maximally pruned graph) are the uniform nodes whose
//      parent is not uniform; in trie DFS order they are already sorted by key.
//   3. LCP[i] = depth of the lowest common ancestor of final nodes i-1, i.  Predecessor labels
//      come from the input graph; out-degrees from suffix links: the successors of key p are the
//      final nodes below trie node p[1..] that have p[0] among their predecessor labels
//      (paper.tex:252, :551-553).
//   4. R[] (redundant pointers) by the in-order traversal rule of src/gcsa.cpp:590-619.
// The Python side (workload/index_arrays.py) turns the table into bitvectors, samples, counters.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

namespace {

constexpr int SIGMA = 7;
constexpr u32 NONE = 0xFFFFFFFFu;

struct Member { u32 v, u; };

struct Trie
{
  std::vector<u32> parent, first_child;
  std::vector<u8> comp, mask, kind;      // kind: 0 internal, 1 leaf
  std::vector<u16> depth;
  std::vector<u64> set_ptr;              // into pool (leaves; propagated to uniform internal nodes)
  std::vector<u32> set_len;
  std::vector<u32> pool;                 // start positions, sorted per set

  u32 add(u32 par, u8 c, u16 d)
  {
    parent.push_back(par); first_child.push_back(NONE); comp.push_back(c); mask.push_back(0);
    kind.push_back(0); depth.push_back(d); set_ptr.push_back(0); set_len.push_back(0);
    return u32(parent.size() - 1);
  }
  size_t size() const { return parent.size(); }
  inline u32 child(u32 node, u32 c) const
  {
    if(kind[node] != 0 || !((mask[node] >> c) & 1)) { return NONE; }
    return first_child[node] + u32(__builtin_popcount(mask[node] & ((1u << c) - 1)));
  }
};

}  // namespace

extern "C" {

struct gcsa_node_table
{
  u64 n, total_vals;
  u8* pred_mask; u32* outdeg; u8* lcp; u64* val_off; u64* vals; u32* redundant; u16* key_len;
  u64 trie_nodes, max_members;
  double seconds;
};

void gcsa_build_free(gcsa_node_table* t)
{
  free(t->pred_mask); free(t->outdeg); free(t->lcp); free(t->val_off); free(t->vals);
  free(t->redundant); free(t->key_len);
  memset(t, 0, sizeof(*t));
}

// Returns 0 on success, < 0 on a violated assumption (message on stderr).
int gcsa_build_nodes(u64 N, const u8* comp, const u64* value, const u64* succ_off, const u32* succ,
                     u64 K, int threads, int verbose, gcsa_node_table* out)
{
  double t0 = omp_get_wtime();
  if(threads <= 0) { threads = omp_get_max_threads(); }
  if(N == 0 || N >= NONE || K == 0 || K > 256) { fprintf(stderr, "gcsa_build_nodes: bad N or K\n"); return -1; }
  memset(out, 0, sizeof(*out));

  // predecessor label mask of every position
  std::vector<u8> predmask(N, 0);
  for(u64 v = 0; v < N; v++)
  {
    if(succ_off[v + 1] == succ_off[v]) { fprintf(stderr, "gcsa_build_nodes: position %lu has no successor\n", (unsigned long)v); return -2; }
    for(u64 j = succ_off[v]; j < succ_off[v + 1]; j++) { predmask[succ[j]] |= u8(1u << comp[v]); }
  }

  Trie T;
  T.add(NONE, 0, 0);  // root

  // ---- phase 1: level-synchronous expansion -------------------------------------------------
  std::vector<u32> act;            // active nodes of the current level
  std::vector<u64> moff;           // member offsets, act.size() + 1
  std::vector<Member> members;

  // level 1 from the root: bucket all positions by comp
  {
    u64 cnt[SIGMA + 1] = {0};
    for(u64 v = 0; v < N; v++) { cnt[comp[v]]++; }
    std::vector<Member> lvl(N);
    u64 base[SIGMA + 1], run = 0;
    for(int c = 0; c < SIGMA; c++) { base[c] = run; run += cnt[c]; }
    u64 fill[SIGMA];
    for(int c = 0; c < SIGMA; c++) { fill[c] = base[c]; }
    for(u64 v = 0; v < N; v++) { lvl[fill[comp[v]]++] = Member{u32(v), u32(v)}; }
    moff.push_back(0);
    for(int c = 0; c < SIGMA; c++)
    {
      if(cnt[c] == 0) { continue; }
      u32 node = T.add(0, u8(c), 1);
      if(T.first_child[0] == NONE) { T.first_child[0] = node; }
      T.mask[0] |= u8(1u << c);
      if(cnt[c] == 1 || K == 1)
      {
        T.kind[node] = 1; T.set_ptr[node] = T.pool.size(); T.set_len[node] = u32(cnt[c]);
        for(u64 j = base[c]; j < base[c] + cnt[c]; j++) { T.pool.push_back(lvl[j].v); }
      }
      else
      {
        act.push_back(node);
        members.insert(members.end(), lvl.begin() + base[c], lvl.begin() + base[c] + cnt[c]);
        moff.push_back(members.size());
      }
    }
  }

  u64 max_members = members.size();
  for(u64 d = 1; !act.empty(); d++)   // expanding nodes of depth d into depth d + 1
  {
    size_t m = act.size();
    if(verbose) { fprintf(stderr, "  depth %lu: %zu active nodes, %zu members, %zu trie nodes\n", (unsigned long)d, m, members.size(), T.size()); }
    std::vector<u32> cnt(m * 8, 0);
    #pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for(size_t i = 0; i < m; i++)
    {
      u32* c = cnt.data() + i * 8;
      for(u64 k = moff[i]; k < moff[i + 1]; k++)
      {
        u32 u = members[k].u;
        for(u64 j = succ_off[u]; j < succ_off[u + 1]; j++) { c[comp[succ[j]]]++; }
      }
    }
    std::vector<u64> nbase(m + 1, 0);
    for(size_t i = 0; i < m; i++)
    {
      u64 s = 0;
      for(int c = 0; c < SIGMA; c++) { s += cnt[i * 8 + c]; }
      nbase[i + 1] = nbase[i] + s;
    }
    std::vector<Member> tmp(nbase[m]);
    std::vector<u32> fsz(m * 8, 0);      // size after dedup
    std::vector<u8> fkind(m * 8, 0);     // 0 absent, 1 leaf, 2 active
    bool last_level = (d + 1 == K);
    #pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for(size_t i = 0; i < m; i++)
    {
      u64 fill[SIGMA], start[SIGMA], run = nbase[i];
      for(int c = 0; c < SIGMA; c++) { start[c] = fill[c] = run; run += cnt[i * 8 + c]; }
      for(u64 k = moff[i]; k < moff[i + 1]; k++)
      {
        Member mb = members[k];
        for(u64 j = succ_off[mb.u]; j < succ_off[mb.u + 1]; j++)
        {
          u32 w = succ[j];
          tmp[fill[comp[w]]++] = Member{mb.v, w};
        }
      }
      for(int c = 0; c < SIGMA; c++)
      {
        u64 b = start[c], e = fill[c];
        if(b == e) { continue; }
        // members are grouped by start v (stable); sort + unique each group by u
        u64 tail = b;
        bool single = true;
        for(u64 g = b; g < e; )
        {
          u64 h = g + 1;
          while(h < e && tmp[h].v == tmp[g].v) { h++; }
          if(h - g > 1)
          {
            std::sort(tmp.begin() + g, tmp.begin() + h, [](const Member& x, const Member& y) { return x.u < y.u; });
          }
          for(u64 k = g; k < h; k++)
          {
            if(k > g && tmp[k].u == tmp[k - 1].u) { continue; }
            tmp[tail++] = tmp[k];
          }
          if(tmp[g].v != tmp[b].v) { single = false; }
          g = h;
        }
        fsz[i * 8 + c] = u32(tail - b);
        fkind[i * 8 + c] = (single || last_level) ? 1 : 2;
      }
    }
    // create the children (serial, in (node, comp) order = lexicographic order of the level)
    std::vector<u32> next_act;
    std::vector<u64> next_moff(1, 0);
    std::vector<u64> src_begin;   // for active children: where their members sit in tmp
    u64 next_total = 0;
    for(size_t i = 0; i < m; i++)
    {
      u32 node = act[i];
      u64 run = nbase[i];
      for(int c = 0; c < SIGMA; c++)
      {
        u64 b = run; run += cnt[i * 8 + c];
        u8 k = fkind[i * 8 + c];
        if(k == 0) { continue; }
        u32 child = T.add(node, u8(c), u16(d + 1));
        if(T.first_child[node] == NONE) { T.first_child[node] = child; }
        T.mask[node] |= u8(1u << c);
        u32 sz = fsz[i * 8 + c];
        if(k == 1)
        {
          T.kind[child] = 1; T.set_ptr[child] = T.pool.size();
          u32 len = 0;
          for(u64 j = b; j < b + sz; j++)
          {
            if(j > b && tmp[j].v == tmp[j - 1].v) { continue; }
            T.pool.push_back(tmp[j].v); len++;
          }
          T.set_len[child] = len;
          // members of one start may be interleaved only by u; v order follows position order
        }
        else
        {
          next_act.push_back(child); src_begin.push_back(b);
          next_total += sz; next_moff.push_back(next_total);
        }
      }
    }
    std::vector<Member> next_members(next_total);
    #pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for(size_t i = 0; i < next_act.size(); i++)
    {
      std::copy(tmp.begin() + src_begin[i], tmp.begin() + src_begin[i] + (next_moff[i + 1] - next_moff[i]),
                next_members.begin() + next_moff[i]);
    }
    act.swap(next_act); moff.swap(next_moff); members.swap(next_members);
    if(members.size() > max_members) { max_members = members.size(); }
    if(T.size() >= NONE - 16) { fprintf(stderr, "gcsa_build_nodes: trie too large\n"); return -3; }
  }
  std::vector<Member>().swap(members);
  size_t TN = T.size();
  if(verbose) { fprintf(stderr, "  trie: %zu nodes, pool %zu (%.1f s)\n", TN, T.pool.size(), omp_get_wtime() - t0); }

  // leaf sets must be sorted for comparisons (they are: members are in start order) -- verify cheaply
  // ---- phase 2: uniformity, bottom-up (children have larger ids than parents) -----------------
  std::vector<u8> uniform(TN, 0);
  for(size_t x = TN; x-- > 0; )
  {
    if(T.kind[x] == 1) { uniform[x] = 1; continue; }
    if(x == 0) { uniform[x] = 0; continue; }
    u32 fc = T.first_child[x];
    u32 nc = u32(__builtin_popcount(T.mask[x]));
    bool ok = true;
    for(u32 j = 0; j < nc && ok; j++)
    {
      u32 ch = fc + j;
      if(!uniform[ch]) { ok = false; break; }
      if(j > 0)
      {
        if(T.set_len[ch] != T.set_len[fc] ||
           memcmp(T.pool.data() + T.set_ptr[ch], T.pool.data() + T.set_ptr[fc], size_t(T.set_len[fc]) * sizeof(u32)) != 0)
        { ok = false; }
      }
    }
    if(ok) { uniform[x] = 1; T.set_ptr[x] = T.set_ptr[fc]; T.set_len[x] = T.set_len[fc]; }
  }

  // ---- phase 3: final nodes, DFS numbering, LCP -------------------------------------------------
  // state: 0 = above the final cut (non-uniform), 1 = final path node, 2 = below a final node
  std::vector<u8> state(TN, 0);
  for(size_t x = 1; x < TN; x++)
  {
    u8 ps = state[T.parent[x]];
    if(ps != 0) { state[x] = 2; }
    else { state[x] = uniform[x] ? 1 : 0; }
  }
  std::vector<u32> leafcount(TN, 0), offset(TN, 0);
  for(size_t x = TN; x-- > 0; )
  {
    if(state[x] == 1) { leafcount[x] = 1; }
    if(state[x] != 2 && x > 0) { leafcount[T.parent[x]] += leafcount[x]; }
  }
  u64 n = leafcount[0];
  if(n == 0 || n >= NONE) { fprintf(stderr, "gcsa_build_nodes: bad node count\n"); return -4; }
  std::vector<u8> lcp(n, 0);
  std::vector<u32> final_node(n, 0);
  for(size_t x = 0; x < TN; x++)
  {
    if(state[x] == 1) { final_node[offset[x]] = u32(x); continue; }
    if(state[x] != 0) { continue; }
    u32 fc = T.first_child[x], nc = u32(__builtin_popcount(T.mask[x]));
    u32 run = offset[x];
    for(u32 j = 0; j < nc; j++)
    {
      offset[fc + j] = run;
      if(j > 0) { lcp[run] = u8(T.depth[x]); }
      run += leafcount[fc + j];
    }
  }

  // suffix links (only meaningful while they stay inside expanded nodes)
  std::vector<u32> sl(TN, NONE);
  sl[0] = 0;
  for(size_t x = 1; x < TN; x++)
  {
    if(state[x] == 2) { continue; }
    u32 p = T.parent[x];
    if(p == 0) { sl[x] = 0; continue; }
    u32 sp = sl[p];
    sl[x] = (sp == NONE ? NONE : T.child(sp, T.comp[x]));
  }

  // ---- per-node outputs ---------------------------------------------------------------------------
  out->n = n;
  out->pred_mask = (u8*)calloc(n, 1);
  out->outdeg = (u32*)calloc(n, sizeof(u32));
  out->lcp = (u8*)calloc(n, 1);
  out->key_len = (u16*)calloc(n, sizeof(u16));
  out->val_off = (u64*)calloc(n + 1, sizeof(u64));
  out->redundant = (u32*)calloc(n > 1 ? n - 1 : 1, sizeof(u32));
  memcpy(out->lcp, lcp.data(), n);
  for(u64 i = 0; i < n; i++) { out->val_off[i + 1] = out->val_off[i] + T.set_len[final_node[i]]; }
  u64 total_vals = out->val_off[n];
  out->total_vals = total_vals;
  out->vals = (u64*)calloc(total_vals > 0 ? total_vals : 1, sizeof(u64));
  int bad = 0;
  #pragma omp parallel for schedule(static) num_threads(threads)
  for(u64 i = 0; i < n; i++)
  {
    u32 x = final_node[i];
    out->key_len[i] = T.depth[x];
    const u32* s = T.pool.data() + T.set_ptr[x];
    u64* dst = out->vals + out->val_off[i];
    u8 pm = 0;
    for(u32 j = 0; j < T.set_len[x]; j++) { pm |= predmask[s[j]]; dst[j] = value[s[j]]; }
    if(T.set_len[x] > 1) { std::sort(dst, dst + T.set_len[x]); }
    out->pred_mask[i] = pm;
  }
  // prefix counts of B_c over final nodes, then out-degrees through suffix links
  std::vector<u32> bc((n + 1) * SIGMA, 0);
  for(u64 i = 0; i < n; i++)
  {
    for(int c = 0; c < SIGMA; c++) { bc[(i + 1) * SIGMA + c] = bc[i * SIGMA + c] + ((out->pred_mask[i] >> c) & 1); }
  }
  #pragma omp parallel for schedule(static) num_threads(threads) reduction(+:bad)
  for(u64 i = 0; i < n; i++)
  {
    u32 x = final_node[i];
    // first character of the key = comp of the depth-1 ancestor
    u32 y = x;
    while(T.depth[y] > 1) { y = T.parent[y]; }
    u32 c = T.comp[y];
    u32 s = sl[x];
    if(s == NONE || state[s] == 2) { bad++; continue; }
    u64 lo = offset[s], hi = lo + leafcount[s];
    out->outdeg[i] = bc[hi * SIGMA + c] - bc[lo * SIGMA + c];
  }
  if(bad) { fprintf(stderr, "gcsa_build_nodes: %d final nodes without a usable suffix link\n", bad); gcsa_build_free(out); return -5; }

  // ---- phase 4: redundant pointers R[0..n-2] (src/gcsa.cpp:590-619, restated) -------------------
  {
    std::vector<u32> prev_occ(N, 0);       // indexed by start position: last node index + 1
    std::vector<u32> node_lcp, first_time, last_time;
    for(u64 i = 0; i < n; i++)
    {
      u32 cur = u32(lcp[i]) + (i > 0 ? 1 : 0);   // LCP[0] acts as -1
      while(!node_lcp.empty() && node_lcp.back() > cur) { node_lcp.pop_back(); first_time.pop_back(); last_time.pop_back(); }
      if(!node_lcp.empty() && node_lcp.back() == cur) { last_time.back() = u32(i); }
      else { node_lcp.push_back(cur); first_time.push_back(u32(i)); last_time.push_back(u32(i)); }
      u32 x = final_node[i];
      const u32* s = T.pool.data() + T.set_ptr[x];
      for(u32 j = 0; j < T.set_len[x]; j++)
      {
        u32 p = prev_occ[s[j]];
        if(p > 0)
        {
          size_t pos = std::lower_bound(last_time.begin(), last_time.end(), p) - last_time.begin();
          out->redundant[first_time[pos] - 1]++;
        }
        prev_occ[s[j]] = u32(i + 1);
      }
    }
  }
  out->trie_nodes = TN; out->max_members = max_members;
  out->seconds = omp_get_wtime() - t0;
  if(verbose) { fprintf(stderr, "  %lu path nodes, %lu values, %.1f s\n", (unsigned long)n, (unsigned long)total_vals, out->seconds); }
  return 0;
}

// int_vector<0> packing helper for large sample arrays (LSB-first, element i at bit i * width).
void gcsa_pack_ints(const u64* values, u64 count, u64 width, u64* out_words)
{
  for(u64 i = 0; i < count; i++)
  {
    u64 pos = i * width, word = pos >> 6, shift = pos & 63;
    out_words[word] |= values[i] << shift;
    if(shift + width > 64) { out_words[word + 1] |= values[i] >> (64 - shift); }
  }
}


// Cyclic "de Bruijn-like" text for footprint-scale indexes that need NO suffix sorting.
// A binary m-sequence of even degree d (maximal-length LFSR, period 2^d - 1) read two bits at a
// time is a cyclic text over {A,C,G,T} of length N = 2^d - 1 in which every d/2-mer except A^(d/2)
// occurs exactly once (the period is odd, so the symbol windows sweep all bit offsets).  The rank
// of rotation i among all rotations is therefore the value of its first d/2 symbols minus one.
//   sym[i]  = symbol i (0..3),   rank[i] = lexicographic rank of the rotation starting at i.
// Returns 0, or -1 if `taps` is not primitive (the state did not return after exactly 2^d - 1 steps).
int gcsa_mseq_text(int degree, const int* taps, int ntaps, u8* sym, u32* rank)
{
  if(degree < 4 || degree > 32 || (degree & 1)) { return -2; }
  const u64 period = (u64(1) << degree) - 1, mask = period;
  // Fibonacci LFSR on the window of the next `degree` output bits (MSB = oldest = first bit of the
  // rotation): a[n + d] = XOR of a[n + d - tap] over the taps.  The step is linear over GF(2).
  auto step = [&](u64 st) -> u64
  {
    u64 fb = 0;
    for(int t = 0; t < ntaps; t++) { fb ^= (st >> (taps[t] - 1)) & 1; }
    return ((st << 1) | fb) & mask;
  };
  // transition matrix A (column c = image of bit c) and its powers A^(2^j), for jump-ahead
  typedef std::vector<u64> Mat;
  auto mat_vec = [&](const Mat& M, u64 v) -> u64
  {
    u64 r = 0;
    while(v) { int c = __builtin_ctzll(v); v &= v - 1; r ^= M[size_t(c)]; }
    return r;
  };
  std::vector<Mat> power(size_t(degree) + 2, Mat(size_t(degree), 0));
  for(int c = 0; c < degree; c++) { power[0][size_t(c)] = step(u64(1) << c); }
  for(size_t j = 1; j < power.size(); j++)
  {
    for(int c = 0; c < degree; c++) { power[j][size_t(c)] = mat_vec(power[j - 1], power[j - 1][size_t(c)]); }
  }
  auto jump = [&](u64 st, u64 steps) -> u64      // A^steps * st
  {
    for(size_t j = 0; steps != 0; j++, steps >>= 1) { if(steps & 1) { st = mat_vec(power[j], st); } }
    return st;
  };
  const u64 start = 1;
  // primitivity: A^period fixes the start state and A^(period / p) does not, for every prime p | period
  if(jump(start, period) != start) { return -1; }
  {
    u64 rest = period;
    for(u64 p = 2; p * p <= rest; p++)
    {
      if(rest % p != 0) { continue; }
      while(rest % p == 0) { rest /= p; }
      if(jump(start, period / p) == start) { return -1; }
    }
    if(rest > 1 && rest != period && jump(start, period / rest) == start) { return -1; }
    if(rest == period && period > 1 && false) { return -1; }
  }
  // symbol i consumes output bits 2i, 2i + 1; the window of rotation i is the state before them
  const int threads = omp_get_max_threads();
  const u64 chunks = u64(threads) * 4, per = (period + chunks - 1) / chunks;
  #pragma omp parallel for schedule(dynamic, 1)
  for(u64 ch = 0; ch < chunks; ch++)
  {
    u64 begin = ch * per, end = begin + per < period ? begin + per : period;
    if(begin >= end) { continue; }
    u64 state = jump(start, 2 * begin);
    for(u64 i = begin; i < end; i++)
    {
      rank[i] = u32(state - 1);
      sym[i] = u8((state >> (degree - 2)) & 3);
      state = step(step(state));
    }
  }
  return 0;
}

// Cyclic text of an LFSR cycle that is NOT maximal: the cycle through state 1 of the recurrence a[n + d] = XOR of
// a[n + d - tap] must have exactly `period` states (odd), e.g. (2^d - 1) / 3 for a suitable irreducible polynomial of even
// degree d.  Symbol i consumes output bits 2i, 2i + 1, so the d/2-mer at position i is the state before them: the
// `period` d/2-mers of the cyclic text are the states of the cycle, all distinct -- but, unlike gcsa_mseq_text, not
// all d/2-mers exist, so ranks are not closed-form values (workload/dbg_torch.py ranks them through a bitmap of the
// k-mer universe).  degree <= 62.  Returns 0, -1 if the cycle length is not `period`, -2 on bad arguments.
int gcsa_lfsr_text(int degree, const int* taps, int ntaps, u64 period, u8* sym)
{
  if(degree < 4 || degree > 62 || (degree & 1) || (period & 1) == 0) { return -2; }
  const u64 mask = (u64(1) << degree) - 1;
  auto step = [&](u64 st) -> u64
  {
    u64 fb = 0;
    for(int t = 0; t < ntaps; t++) { fb ^= (st >> (taps[t] - 1)) & 1; }
    return ((st << 1) | fb) & mask;
  };
  typedef std::vector<u64> Mat;
  auto mat_vec = [&](const Mat& M, u64 v) -> u64
  {
    u64 r = 0;
    while(v) { int c = __builtin_ctzll(v); v &= v - 1; r ^= M[size_t(c)]; }
    return r;
  };
  std::vector<Mat> power(size_t(degree) + 2, Mat(size_t(degree), 0));
  for(int c = 0; c < degree; c++) { power[0][size_t(c)] = step(u64(1) << c); }
  for(size_t j = 1; j < power.size(); j++)
  {
    for(int c = 0; c < degree; c++) { power[j][size_t(c)] = mat_vec(power[j - 1], power[j - 1][size_t(c)]); }
  }
  auto jump = [&](u64 st, u64 steps) -> u64
  {
    for(size_t j = 0; steps != 0; j++, steps >>= 1) { if(steps & 1) { st = mat_vec(power[j], st); } }
    return st;
  };
  const u64 start = 1;
  if(jump(start, period) != start) { return -1; }
  {
    u64 rest = period;
    for(u64 p = 3; p * p <= rest; p += 2)
    {
      if(rest % p != 0) { continue; }
      while(rest % p == 0) { rest /= p; }
      if(jump(start, period / p) == start) { return -1; }
    }
    if(rest > 1 && rest != period && jump(start, period / rest) == start) { return -1; }
  }
  const int threads = omp_get_max_threads();
  const u64 chunks = u64(threads) * 4, per = (period + chunks - 1) / chunks;
  #pragma omp parallel for schedule(dynamic, 1)
  for(u64 ch = 0; ch < chunks; ch++)
  {
    u64 begin = ch * per, end = begin + per < period ? begin + per : period;
    if(begin >= end) { continue; }
    u64 state = jump(start, 2 * begin);
    for(u64 i = begin; i < end; i++)
    {
      sym[i] = u8((state >> (degree - 2)) & 3);
      state = step(step(state));
    }
  }
  return 0;
}

}  // extern "C"
