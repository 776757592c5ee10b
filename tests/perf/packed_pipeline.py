#!/usr/bin/env python3
"""Host-memory k-mer batches through the chunked pipeline by configuration: lanes x patterns per chunk, 2-bit codes
(gcsa2_find_batch_packed) and bytes (gcsa2_find_batch), pageable and page-locked, on a random 2^LOG-base text (a linear graph:
workload/linear_torch.py; find-only image).  One JSON line per configuration; the knobs are read when the index is created, so
every configuration opens its own handle.

    python tests/perf/packed_pipeline.py [--log 28] [--queries 10000000] [--configs 12:17,12:18,8:18,...]  (lanes:log2 chunk)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", type=int, default=28)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--configs", default="12:17,12:18,12:19,8:17,8:18,8:19,6:19,4:19,16:17,16:18,4:20,6:20")
    ap.add_argument("--extra-env", default="", help="NAME=VALUE[,NAME=VALUE] set for every configuration")
    ap.add_argument("--bytes-too", action="store_true")
    args = ap.parse_args()
    import torch
    from workload import linear_torch
    from gcsa2_amd.binding import GCSA
    dev = torch.device("cuda", 0)
    n, m, nq = 1 << args.log, 32, args.queries
    ix = linear_torch.build_linear(n, 0x6C5A0200, device=dev, with_lcp=False, with_samples=False)
    pats = linear_torch.substring_patterns_torch(n, 0x6C5A0200, nq, m, 0x6C5A0201, device=dev)
    flat = np.ascontiguousarray(pats.reshape(-1))
    off = (np.arange(nq + 1, dtype=np.uint64) * np.uint64(m))
    lut = np.zeros(256, dtype=np.uint64)
    for ch, c in zip(b"ACGT", range(4)):
        lut[ch] = c
    comps = lut[pats]
    codes = np.zeros(nq, dtype=np.uint64)
    for t in range(m):
        codes |= comps[:, m - 1 - t] << np.uint64(2 * t)
    codes = codes.reshape(nq, 1)
    del comps
    p_codes = torch.empty((nq, 1), dtype=torch.int64).pin_memory(); p_codes.numpy().view(np.uint64)[:] = codes
    p_out = torch.empty((nq, 2), dtype=torch.int64).pin_memory()
    out = np.zeros((nq, 2), dtype=np.uint64)
    for kv in filter(None, args.extra_env.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    want = None
    for cfg in args.configs.split(","):
        lanes, chunk = (int(x) for x in cfg.split(":"))
        os.environ["GCSA2_PIPE_LANES"] = str(lanes)
        os.environ["GCSA2_PIPE_CHUNK"] = str(chunk)
        gpu = GCSA(ix, device=0, with_samples=False, with_counters=False, with_lcp=False)
        row = {"lanes": lanes, "chunk_log2": chunk, "extra": args.extra_env}
        legs = [("packed", lambda o: gpu.find_batch_packed(codes, m, out=o), out),
                ("packed_pinned", lambda o: gpu.find_batch_packed(p_codes.numpy().view(np.uint64), m, out=o), p_out.numpy().view(np.uint64))]
        if args.bytes_too:
            legs.append(("bytes", lambda o: gpu.find_batch(flat, off, out=o), out))
        for name, call, buf in legs:
            call(buf)
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                call(buf)
                times.append(time.perf_counter() - t0)
            if want is None:
                want = buf.copy()
            row[name + "_Gqps"] = round(nq / min(times) / 1e9, 3)
            row[name + "_ms"] = [round(t * 1e3, 2) for t in times]
            row[name + "_same"] = bool(np.array_equal(buf, want))
        print(json.dumps(row), flush=True)
        gpu.close()


if __name__ == "__main__":
    main()
