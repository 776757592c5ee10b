#!/usr/bin/env python3
"""The wire-format kernels of the multi-GPU step by themselves: gcsa2_pack_ranges40_device / 32 and their inverses on the
shard of one rank at N = 8 (12.5 M ranges) and on the whole batch (100 M, what the root unpacks).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gcsa2_amd import binding
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    out = {}
    for nq in (12_500_000, 100_000_000):
        sp = torch.randint(0, 5_726_623_061, (nq,), dtype=torch.int64, device=dev)
        rng = torch.stack([sp, sp + torch.randint(0, 3, (nq,), dtype=torch.int64, device=dev) - 1], dim=1).contiguous()
        w40 = torch.zeros(nq * 10 + 16, dtype=torch.uint8, device=dev)
        w32 = torch.zeros(nq * 8, dtype=torch.uint8, device=dev)
        back = torch.zeros_like(rng)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps=10):
            fn(); torch.cuda.synchronize()
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        row = {}
        row["pack40_ms"] = timed(lambda: binding.pack_ranges40_device(rng.data_ptr(), nq, w40.data_ptr(), st.cuda_stream))
        row["unpack40_ms"] = timed(lambda: binding.unpack_ranges40_device(w40.data_ptr(), nq, back.data_ptr(), st.cuda_stream))
        row["roundtrip40_exact"] = bool(torch.equal(back, rng))
        small = rng % (1 << 31)
        small[:, 1] = small[:, 0] + (rng[:, 1] - rng[:, 0])
        row["pack32_ms"] = timed(lambda: binding.pack_ranges32_device(small.data_ptr(), nq, w32.data_ptr(), st.cuda_stream))
        row["unpack32_ms"] = timed(lambda: binding.unpack_ranges32_device(w32.data_ptr(), nq, back.data_ptr(), st.cuda_stream))
        row["roundtrip32_exact"] = bool(torch.equal(back, small))
        row["pack40_GBps"] = nq * 26 / row["pack40_ms"] / 1e6
        out[str(nq)] = row
        del sp, rng, w40, w32, back, small
    print(json.dumps(out))


if __name__ == "__main__":
    main()
