"""The halo schedule of the C-ABI (mpg_halo_plan_host, csrc/mpg_comm.hip) against the torch.distributed form it restates
(dist.HaloSchedule.build), world sizes 2, 3 and 5, on the CPU: banded numbering (range form, both ownership rules), a rank
that needs nothing, and arbitrary numbering (compact form)."""
import threading

import numpy as np
import pytest


def _python_schedules(lists, n_cells, ownership):
    """HaloSchedule.build of every rank, the ranks running in lock-step threads behind a barrier-backed all_gather_object."""
    from mpassit_amd import dist
    world = len(lists)
    barrier = threading.Barrier(world)
    slots = [None] * world
    out = [None] * world

    def run(rank):
        def gather(obj):
            slots[rank] = obj
            barrier.wait()
            res = list(slots)
            barrier.wait()
            return res
        out[rank] = dist.HaloSchedule.build(lists[rank], n_cells, rank, world, gather, ownership=ownership)
    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


def _cases():
    rng = np.random.default_rng(7)
    n = 5000
    for world in (2, 3, 5):
        # banded: rank r needs a contiguous-ish band, neighbours overlap by a strip
        bands = []
        for r in range(world):
            lo, hi = max(0, r * n // world - 120), min(n, (r + 1) * n // world + 90)
            ids = np.arange(lo, hi)
            bands.append(ids[rng.random(ids.size) > 0.05])            # a few cells unreferenced: still "range"
        yield "banded%d" % world, bands, n
        hole = [b.copy() for b in bands]
        hole[world // 2] = np.empty(0, np.int64)                       # a row block outside the mesh footprint
        yield "empty%d" % world, hole, n
        yield "scattered%d" % world, [np.sort(rng.choice(n, 700, replace=False)) for _ in range(world)], n


@pytest.mark.parametrize("ownership", ["aligned", "para_range"])
def test_c_schedule_equals_the_torch_distributed_one(ownership):
    from mpassit_amd import comm
    for name, lists, n in _cases():
        want = _python_schedules(lists, n, ownership)
        for rank, s in enumerate(want):
            got = comm.plan_host(rank, lists, n, ownership)
            what = (name, ownership, rank)
            assert got["mode"] == s.mode, what
            assert got["n_local"] == s.n_local and got["own"] == tuple(s.own), what
            send_n, recv_n = s.counts()
            assert got["send_count"] == send_n, what
            assert [b - a for a, b in got["recv_pos"]] == recv_n, what
            assert [p for p, c in zip(got["recv_pos"], recv_n) if c] == [tuple(p) for p, c in zip(s.recv_pos, recv_n) if c], what
            if s.mode == "range":
                assert got["base"] == s.base and got["own_pos"] == tuple(s.own_pos), what
                assert [a for a, c in zip(got["send_a"], send_n) if c] == [p[0] for p, c in zip(s.send_ids, send_n) if c], what
            else:
                for q in range(len(lists)):
                    assert np.array_equal(got["send_ids"][q], s.send_ids[q]), what + (q,)
