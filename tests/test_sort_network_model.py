"""A model of the register sort for three and six values per lane (kernels_locate.hpp: wave_sort_blocked32_odd) in plain Python:
the two in-lane sorting networks by the 0-1 principle, the whole network -- element R lane + r, mirror step, lane halves, in-lane
bitonic finish -- on random inputs with and without duplicates for 2 .. 64 lanes.  The device code is a transcription of this
model (same comparator lists, same partner rules); the GPU parity tests sort real buckets with it."""
import itertools
import random

import pytest

NET3 = [(0, 1), (1, 2), (0, 1)]
NET6 = [(0, 5), (1, 3), (2, 4), (1, 2), (3, 4), (0, 3), (2, 5), (0, 1), (2, 3), (4, 5), (1, 2), (3, 4)]


def run_net(v, net, base=0):
    for i, j in net:
        if v[base + i] > v[base + j]:
            v[base + i], v[base + j] = v[base + j], v[base + i]


@pytest.mark.parametrize("n,net", [(3, NET3), (6, NET6)])
def test_in_lane_networks_sort_every_zero_one_input(n, net):
    for bits in itertools.product((0, 1), repeat=n):
        v = list(bits)
        run_net(v, net)
        assert v == sorted(v)


def bitonic_finish(v, R):
    if R == 6:
        run_net(v, [(0, 3), (1, 4), (2, 5)])
        run_net(v, NET3, 0)
        run_net(v, NET3, 3)
    else:
        run_net(v, NET3)


def blocked_sort(values, R, lanes):
    V = [values[l * R:(l + 1) * R] for l in range(lanes)]
    for row in V:
        run_net(row, NET3 if R == 3 else NET6)
    M = 1
    while M < lanes:
        new = [row[:] for row in V]
        for l in range(lanes):                                   # the mirror step: lane ^ M, register R - 1 - r
            lower = (l & ((M + 1) // 2)) == 0
            for r in range(R):
                other = V[l ^ M][R - 1 - r]
                new[l][r] = min(V[l][r], other) if lower else max(V[l][r], other)
        V = new
        L = (M + 1) // 4
        while L >= 1:                                            # lane with lane ^ L
            new = [row[:] for row in V]
            for l in range(lanes):
                for r in range(R):
                    new[l][r] = min(V[l][r], V[l ^ L][r]) if (l & L) == 0 else max(V[l][r], V[l ^ L][r])
            V = new
            L //= 2
        for row in V:
            bitonic_finish(row, R)
        M = 2 * M + 1
    return [x for row in V for x in row]


@pytest.mark.parametrize("R", [3, 6])
@pytest.mark.parametrize("lanes", [2, 4, 8, 16, 64])
def test_blocked_network_sorts(R, lanes):
    rng = random.Random(0x6C5A + 64 * R + lanes)
    n = R * lanes
    for trial in range(60):
        values = [rng.randrange(0, 40) for _ in range(n)] if trial % 2 else rng.sample(range(3 * n), n)
        assert blocked_sort(values[:], R, lanes) == sorted(values)
    for bits in itertools.islice(itertools.product((0, 1), repeat=n), 0, 4096):      # (the first inputs of the 0-1 principle; all of them for n = 6, 12)
        assert blocked_sort(list(bits), R, lanes) == sorted(bits)
