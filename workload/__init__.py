"""Synthetic workloads for the GCSA2 query hot path (test / bench infrastructure).

Nothing in here is part of the shipped product path: it only manufactures the
*inputs* (a valid GCSA index as plain arrays, and query patterns) that the
oracle and the HIP engine are both run on.  No real `.gcsa` file can reach the
GPU box, so every index is derived from a seed (SURVEY.md §8(d) "Real data
caveat").

  rng.py            splitmix64, shared by Python, torch and the C++ builder
  graphs.py         seeded input graphs (linear, SNP bubbles, small random graphs, the paper's figure)
  index_arrays.py   path-node table -> the members of gcsa::GCSA / LCPArray as plain arrays
  brute_builder.py  definitional (exponential) order-K maximally pruned de Bruijn graph -> path-node table
  builder.cpp/.py   scalable trie-refinement builder producing the same table (cross-checked in tests)
  linear_torch.py   footprint-scale linear-graph index by prefix doubling with torch (GPU when present)
  patterns.py       seeded query sets (walks through the graph "S", uniform random "U")
  sdsl_format.py    writer of .gcsa / .lcp byte streams (restated SDSL encodings) for the file-reader tests
  cache.py          .npz save / load of an index (one build per node in multi-rank runs)
"""
