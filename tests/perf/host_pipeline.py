#!/usr/bin/env python3
"""Host-memory find() batches by pipeline configuration (lanes, one or two streams per lane), from pageable and from
page-locked memory, on the chr22-like index: `python tests/perf/host_pipeline.py` prints one JSON line per configuration.
The knobs are read when the index is created, so every configuration opens its own handle."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from workload import graphs, builder, patterns
    from gcsa2_amd.binding import open_index
    g = graphs.snp_graph(1 << 22, 0x6C5A0010, 0x6C5A0011)
    ix = builder.build(g, 256, keep_table=False)
    nq, m = 10_000_000, 32
    pats = patterns.walk_patterns(g, nq, m, 0x6C5A0012)
    flat, off = patterns.as_batch(pats)
    p_flat = torch.empty(nq * m, dtype=torch.uint8).pin_memory(); p_flat.numpy()[:] = flat
    p_off = torch.empty(nq + 1, dtype=torch.int64).pin_memory(); p_off.numpy().view(np.uint64)[:] = off
    p_out = torch.empty((nq, 2), dtype=torch.int64).pin_memory()
    out = np.zeros((nq, 2), dtype=np.uint64)
    want = None
    configs = [(int(a), int(b)) for a, b in (c.split(":") for c in (sys.argv[1] if len(sys.argv) > 1 else "12:0,12:1,6:0,6:1,4:1,3:1,8:0,16:0").split(","))]
    for lanes, split in configs:
        os.environ["GCSA2_PIPE_LANES"] = str(lanes)
        os.environ["GCSA2_PIPE_SPLIT"] = str(split)
        gpu, _ = open_index(ix)
        row = {"lanes": lanes, "split": split}
        for name, a, b, c in (("pageable", flat, off, out), ("page_locked", p_flat.numpy(), p_off.numpy().view(np.uint64), p_out.numpy().view(np.uint64))):
            gpu.find_batch(a, b, out=c)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                gpu.find_batch(a, b, out=c)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            if want is None:
                want = c.copy()
            row[name + "_Gqps"] = round(nq / best / 1e9, 3)
            row[name + "_same"] = bool(np.array_equal(c, want))
        print(json.dumps(row), flush=True)
        gpu.close()


if __name__ == "__main__":
    main()
