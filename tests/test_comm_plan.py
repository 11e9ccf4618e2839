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


def _lockstep(world, fn):
    """fn(rank, all_gather_object) for every rank in lock-step threads."""
    barrier = threading.Barrier(world)
    slots, out = [None] * world, [None] * world

    def run(rank):
        def gather(obj):
            slots[rank] = obj
            barrier.wait()
            res = list(slots)
            barrier.wait()
            return res
        try:
            out[rank] = fn(rank, gather)
        except Exception as e:      # noqa: BLE001 -- handed to the caller (a rank that fails AFTER its collectives must not strand the others)
            out[rank] = e
    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


@pytest.mark.parametrize("partition", ["need", "random"])
def test_owned_form_c_schedule_equals_the_torch_distributed_one(partition):
    """Round 5: the caller's own partition of the cells (mpg_halo_build_owned).  `need`: every cell to the lowest rank that references it
    (dist.HaloSchedule.owned_by_need); `random`: an arbitrary assignment, as a coupled model's graph partition would look to this
    library.  C plan == Python plan, world 2 / 3 / 5; the schedules of two ranks agree (what r sends q is what q expects from r, in
    the same order); a cell nobody owns is refused by both."""
    from mpassit_amd import _lib, comm, dist
    rng = np.random.default_rng(11)
    n = 5000
    for world in (2, 3, 5):
        needed = [np.sort(rng.choice(n, 900, replace=False)).astype(np.int32) for _ in range(world)]
        if partition == "need":
            owned = _lockstep(world, lambda r, ag: dist.HaloSchedule.owned_by_need(needed[r], r, world, ag))
            assert sum(o.size for o in owned) == np.unique(np.concatenate(needed)).size              # every referenced cell exactly once
        else:
            owner = rng.integers(0, world, n)
            owned = [np.nonzero(owner == r)[0].astype(np.int32) for r in range(world)]
        py = _lockstep(world, lambda r, ag: dist.HaloSchedule.build_owned(needed[r], owned[r], r, world, ag))
        cs = [comm.plan_owned_host(r, needed, owned) for r in range(world)]
        for r in range(world):
            assert cs[r]["n_local"] == py[r].n_local == needed[r].size
            for q in range(world):
                assert np.array_equal(cs[r]["send_ids"][q], py[r].send_ids[q]) and np.array_equal(cs[r]["recv_ids"][q], py[r].recv_pos[q]), (world, r, q)
                # the two sides of one transfer name the same cells in the same order
                assert np.array_equal(owned[r][cs[r]["send_ids"][q]], needed[q][cs[q]["recv_ids"][r]]), (world, r, q)
            got = np.sort(np.concatenate(cs[r]["recv_ids"]))
            assert np.array_equal(got, np.arange(needed[r].size))                                       # every needed cell arrives exactly once
        if partition == "need":                                                                        # traffic: only what row blocks share
            sent = sum(cs[r]["send_ids"][q].size for r in range(world) for q in range(world) if q != r)
            assert sent == sum(x.size for x in needed) - np.unique(np.concatenate(needed)).size
    # a referenced cell that nobody owns
    needed = [np.array([1, 5, 9], np.int32), np.array([2, 5], np.int32)]
    owned = [np.array([1, 5], np.int32), np.array([2], np.int32)]
    with pytest.raises(_lib.MpgError, match="no rank owns"):
        comm.plan_owned_host(0, needed, owned)
    res = _lockstep(2, lambda r, ag: dist.HaloSchedule.build_owned(needed[r], owned[r], r, 2, ag))
    assert all(isinstance(x, ValueError) and "rank 0's rows reference cell 9, which no rank owns" in str(x) for x in res)   # all ranks refuse together
    assert comm.plan_owned_host(1, needed, owned)["n_local"] == 2        # (the plan of one rank alone, without the collective check)
    # a cell with two owners is no partition: the torch.distributed schedule refuses it on every rank together, as mpg_halo_build_owned does
    # (round-5 advisor: the two transports must accept the same inputs)
    owned2 = [np.array([1, 5, 9], np.int32), np.array([2, 5], np.int32)]
    res = _lockstep(2, lambda r, ag: dist.HaloSchedule.build_owned(needed[r], owned2[r], r, 2, ag))
    assert all(isinstance(x, ValueError) and "cell 5 is owned by ranks [0, 1]" in str(x) for x in res)


# ---- the id-file acceptance rule of mpg_comm_init (pure host logic, no GPU, no RCCL) --------------------------------------
def _idfile(magic=b"MPGRCCL2", tag=0, written_ns=0, nranks=2, size=None):
    import struct
    b = struct.pack("<8sQqii", magic, tag, written_ns, nranks, 0) + bytes(128)          # ncclUniqueId: 128 opaque bytes
    return b if size is None else b[:size]


def test_stale_or_foreign_id_files_are_refused():
    """A reader accepts only THIS launch's id file: right size / magic / nranks / launch tag, and not written long before
    the reader itself started (a file left behind by a killed launch made ncclCommInitRank hang for ever)."""
    import ctypes as C

    from mpassit_amd import _lib
    lib = _lib.load()

    def verdict(b, nranks=2, tag=0, loaded_ns=10 ** 12, stale_s=300.0):
        r = lib.mpg_comm_idfile_verdict(b, C.c_int64(len(b)), C.c_int(nranks), C.c_uint64(tag), C.c_int64(loaded_ns), C.c_double(stale_s))
        return None if r is None else r.decode()

    now = 10 ** 12
    assert verdict(_idfile(written_ns=now)) is None
    assert verdict(_idfile(written_ns=now - 299 * 10 ** 9)) is None                        # rank 0 was up a few minutes earlier
    assert verdict(_idfile(written_ns=now + 5 * 10 ** 9)) is None                          # rank 0 arrives after this reader
    assert "older" in verdict(_idfile(written_ns=now - 301 * 10 ** 9))                     # left behind by an earlier launch
    assert "size" in verdict(_idfile(written_ns=now, size=128))                            # the round-3 format: a bare unique id
    assert "magic" in verdict(_idfile(magic=b"MPGRCCL1", written_ns=now))
    assert "ranks" in verdict(_idfile(written_ns=now, nranks=4))
    assert "launch" in verdict(_idfile(written_ns=now, tag=7), tag=9)
    assert verdict(_idfile(written_ns=now, tag=9), tag=9) is None


def test_world8_schedule_on_configuration_4_moves_what_design_says():
    """DESIGN s5: with the rows of the 1800 x 1060 grid split over 8 ranks and banded cell numbering only the strip of
    cells along a row-block boundary travels -- about 1.5 lattice rows, ~18 MB per neighbour for a batch of 13 fields x
    55 levels of float64.  The needed sets come from the geometry (every cell within one lattice spacing of the block's
    rows, through the projection); the schedule from mpg_halo_plan_host, world 8."""
    from mpassit_amd import comm, dist, synth, workloads
    g = workloads.conus_lambert_grid()
    m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000)
    ci, cj = g.proj.latlon_to_ij(np.degrees(m.latCell), np.degrees(m.lonCell))               # mass point (i, j) = (1, 1) first
    spacing = ((1800 * 1.1) * (1060 * 1.1) / (3.0e6 * (3 ** 0.5 / 2))) ** 0.5                  # synth.regional_mesh_for_lambert
    world, lists = 8, []
    for r in range(world):
        j0, j1 = dist.row_block(g.ny, world, r)
        keep = (cj >= j0 + 1 - spacing) & (cj <= j1 + spacing) & (ci >= 1 - spacing) & (ci <= g.nx + spacing)
        lists.append(np.flatnonzero(keep))
    rows, es = 13 * 55, 8
    for r in range(world):
        p = comm.plan_host(r, lists, m.nCells, "aligned")
        assert p["mode"] == "range"
        recv = [b - a for a, b in p["recv_pos"]]
        talk = [q for q in range(world) if recv[q] or p["send_count"][q]]
        assert set(talk) <= {r - 1, r + 1}                                                    # row-block neighbours only
        for q in talk:
            mb = recv[q] * rows * es / 1e6
            assert 4.0 < mb < 30.0, (r, q, mb)             # half / one / one and a half lattice rows of 2 101 cells: 5.5 / 11.5 / 17.5 MB
        # a rank holds its own eighth plus the two strips, not the mesh
        assert p["n_local"] < 1.15 * m.nCells / world + 4 * 2200
    # para_range ownership (the reference's equal blocks) on the same needs: far more travels
    # (the outer ranks: the mesh's margin shifts their equal block away from their rows; the middle ranks nearly coincide)
    worst = max(sum(b - a for a, b in comm.plan_host(r, lists, m.nCells, "para_range")["recv_pos"]) for r in range(world))
    assert worst * rows * es / 1e6 > 100.0
