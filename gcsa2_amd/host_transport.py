"""A gather through host memory over torch.distributed, for `Comm.custom` (gcsa2_comm_create_custom): the transport of
hosts without RCCL, and of world-size-2 / 3 runs on ONE GPU, where RCCL refuses ranks that share a device.  Everything above
the transport -- shards, the three gathers of the matching statistics, the totals / offsets / values exchange of locate()
with its CSR rebasing, the failure protocol -- is the library's C++ as with RCCL; only the bytes travel differently:
device -> host, `dist.send` / `dist.recv` of byte tensors (any backend that moves CPU tensors: gloo), host -> device."""
import ctypes as C


def _hip():
    """The HIP runtime already mapped into this process (torch's, which the engine shares)."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    return C.CDLL("libamdhip64.so")


class HostGather:
    H2D, D2H, D2D = 1, 2, 3            # hipMemcpyKind

    def __init__(self, dist, rank: int, world: int):
        self.dist, self.rank, self.world = dist, rank, world
        self.hip = _hip()
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0
        self.bytes_moved = 0

    def _check(self, code, what):
        if code != 0:
            raise RuntimeError(f"{what} failed with hipError {code}")

    def __call__(self, d_send, sizes, d_recv, root, stream):
        import torch
        self.calls += 1
        self._check(self.hip.hipStreamSynchronize(stream or None), "hipStreamSynchronize")      # what we send was produced on `stream`
        if self.rank == root:
            at = 0
            for r in range(self.world):
                n = sizes[r]
                if n > 0 and r == root:
                    if d_recv + at != d_send:
                        self._check(self.hip.hipMemcpy(d_recv + at, d_send, n, self.D2D), "hipMemcpy")
                elif n > 0:
                    buf = torch.empty(n, dtype=torch.uint8)
                    self.dist.recv(buf, src=r)
                    self._check(self.hip.hipMemcpy(d_recv + at, buf.data_ptr(), n, self.H2D), "hipMemcpy")
                    self.bytes_moved += n
                at += n
        elif sizes[self.rank] > 0:
            n = sizes[self.rank]
            buf = torch.empty(n, dtype=torch.uint8)
            self._check(self.hip.hipMemcpy(buf.data_ptr(), d_send, n, self.D2H), "hipMemcpy")
            self.dist.send(buf, dst=root)
            self.bytes_moved += n
        return 0
