"""Save / load IndexArrays as an uncompressed .npz (so that only one rank of a node builds)."""
import numpy as np

from .index_arrays import IndexArrays

_SCALARS = ["n", "e", "order", "sigma", "fast_chars", "sample_count", "sample_width",
            "extra_values_len", "redundant_len", "lcp_size", "lcp_branching"]
_ARRAYS = ["char2comp", "C", "edges", "sampled_paths", "stored_samples", "stored_samples_plain",
           "samples", "extra_filter", "extra_values", "redundant", "lcp_offsets", "lcp_data"]


def save(path, ix: IndexArrays):
    d = {k: np.asarray(getattr(ix, k)) for k in _ARRAYS}
    d["scalars"] = np.array([int(getattr(ix, k)) for k in _SCALARS], dtype=np.uint64)
    for c in range(ix.sigma):
        d[f"bwt{c}"] = ix.bwt[c]
    np.savez(path, **d)


def load(path) -> IndexArrays:
    z = np.load(path)
    sc = {k: int(v) for k, v in zip(_SCALARS, z["scalars"])}
    arrays = {k: z[k] for k in _ARRAYS}
    bwt = [z[f"bwt{c}"] for c in range(sc["sigma"])]
    return IndexArrays(bwt=bwt, table=None, **sc, **arrays)
