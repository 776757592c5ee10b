"""Synthetic workloads for the GCSA2 query hot path (test / bench infrastructure).

Nothing in here is part of the shipped product path: it only manufactures the
*inputs* (a valid GCSA index as plain arrays, and query patterns) that the
oracle and the HIP engine are both run on.  No real `.gcsa` file can reach the
GPU box, so every index is derived from a seed (SURVEY.md §8(d) "Real data
caveat").

  rng.py            splitmix64, shared by Python and the C++ builder
  graphs.py         seeded input graphs (linear, SNP bubbles, small random DAGs, the paper's figure)
  brute_builder.py  definitional (exponential) order-K maximally pruned de Bruijn graph -> GCSA arrays
  builder.cpp       scalable trie-refinement builder producing the same arrays (see builder.py)
  patterns.py       seeded query sets (substring walks "S", uniform random "U")
"""
