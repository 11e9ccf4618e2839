"""Multi-GPU path: target rows sharded over ranks, source-cell halo exchanged over RCCL/xGMI.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
Decomposition mirrors the reference's two data-parallel splits:
  * target grid split along j into contiguous row blocks (regDecomp=(/1,npets/), model_grid.F90:693),
  * source cells split into contiguous id blocks by `para_range` (model_grid.F90:423-438,2428-2441).
Every target point is owned by exactly one rank, so there is no reduction collective; the only data
exchange is the source halo (ESMF hides it in the route handle): one all-to-all-v per
field batch.  The schedule is computed once per handle ("geometry halo"), the values move per batch.

Two halo forms:
  range   : cell numbering is spatially banded (MPAS regional meshes ordered by rows, the synthetic
            bench meshes): a rank's local source space is the contiguous global id range [g0, g1) that
            covers its own block plus everything its rows reference; own data lives in place and only
            the two thin side strips travel.
  compact : arbitrary numbering: local source space = sorted unique referenced ids; every referenced
            value is gathered (mpg_pack_dev) and sent, including the rank's own.
"""
from dataclasses import dataclass, field

import numpy as np


def para_range(n1, n2, nprocs, irank):
    """model_grid.F90:2428-2441: inclusive 1-based block [ista, iend] of rank irank."""
    iwork1, iwork2 = divmod(n2 - n1 + 1, nprocs)
    ista = irank * iwork1 + n1 + min(irank, iwork2)
    iend = ista + iwork1 - 1
    if iwork2 > irank:
        iend += 1
    return ista, iend


def row_quantum(nx, ny, world):
    """Rows per alignment quantum of a row split of an nx-wide grid, or 1: a rank's result is [nlev][rows][nx], and level k's plane
    starts on a 128-byte line for every k only when rows * nx is a multiple of 32 points (float32; 16 for float64) -- the Regrid
    kernels' stores of a plane that does not are partial lines at both ends of every run (csrc/geom.h stream_nt,
    profiles/r06_plane_alignment.md: 13 % on float32 results).  Block boundaries on multiples of q = 32 / gcd(nx, 32) rows give every
    rank but possibly the last such planes; taken only where half a quantum is at most 3 % of a block (nx = 1800: q = 4; an odd nx
    would need q = 32: the split stays para_range's)."""
    from math import gcd
    q = 32 // gcd(int(nx), 32)
    return q if q > 1 and 16 * q * world <= ny else 1


def row_block(ny, world, rank, nx=None):
    """0-based half-open mass-row block [j0, j1) of `rank`: para_range's split (model_grid.F90:2428-2441), or -- given the grid's
    width -- the same blocks with their boundaries rounded to whole alignment quanta (row_quantum)."""
    q = row_quantum(nx, ny, world) if nx else 1
    if q > 1:
        bnd = lambda r: ny if r >= world else q * ((2 * r * ny + world * q) // (2 * world * q))   # noqa: E731
        return bnd(rank), bnd(rank + 1)
    a, b = para_range(1, ny, world, rank)
    return a - 1, b


def cell_block(n_cells, world, rank):
    a, b = para_range(1, n_cells, world, rank)
    return a - 1, b


def read_block_decomp_file(path, n_cells, nranks, rank):
    """`block_decomp_file` of the namelist (program_setup.F90:38,148-152; read_block_decomp_file, model_grid.F90:2367-2426): an MPAS graph
    partition file (`x1.655362.graph.info.part.8`) -- one line per cell with the number of the PET that owns it, blank lines skipped.
    The reference checks that the file lists exactly nCells cells and was made for exactly npets processes, and takes the cells whose
    number is its own rank (`elemIDs`, model_grid.F90:437).  -> this rank's sorted 0-based cell ids: the `owned_ids` of
    mpg_halo_build_owned / HaloSchedule.build_owned."""
    import os
    if not os.path.exists(path):
        raise FileNotFoundError("BLOCK DECOMP FILE DOES NOT EXIST: %s" % path)
    with open(path) as f:
        proc = np.array([int(ln.split()[0]) for ln in f if ln.strip()], dtype=np.int64)
    if proc.size != n_cells:
        raise ValueError("BLOCK DECOMPOSITION FILE CONTAINS MORE CELLS THAN INPUT GRID (%d lines, %d cells)" % (proc.size, n_cells))
    if proc.size and int(proc.max()) + 1 != nranks:
        raise ValueError("BLOCK DECOMPOSITION FILE GENERATED FOR %d PROCESSES BUT %d PROCESSORS USED." % (int(proc.max()) + 1, nranks))
    return np.nonzero(proc == rank)[0].astype(np.int32)


@dataclass
class HaloSchedule:
    """Who sends which source cells to whom.  Pure host logic + one torch.distributed exchange;
    device-agnostic (CPU tensors under gloo, CUDA tensors under nccl)."""
    rank: int
    world: int
    mode: str                      # "range" | "compact" | "owned"
    n_local: int                   # length of the local source index space
    own: tuple                     # global [c0, c1) owned by this rank
    base: int = 0                  # range mode: global id of local index 0
    send_ids: list = field(default_factory=list)   # per peer: owned-slab offsets to send (np.int32) or (a, b) range
    recv_pos: list = field(default_factory=list)   # per peer: (a, b) destination range in the local index space
    own_pos: tuple = (0, 0)        # range mode: where the own block sits in the local index space
    owned_ids: object = None       # owned mode: this rank's sorted global cell ids (its rows of `own_rows`, in this order)
    _plans: dict = field(default_factory=dict, repr=False)   # (rows, dtype, device) -> preallocated exchange buffers

    @staticmethod
    def build(needed_ids, n_cells, rank, world, all_gather_object, ownership="aligned", range_slack=1.25):
        """needed_ids: sorted unique global source ids this rank's rows reference.

        ownership = "para_range": source cells owned in equal contiguous id blocks exactly like the reference's
        default partition (model_grid.F90:427); the exchange then moves whatever the geometry dictates.
        ownership = "aligned" (default, banded numbering only): the id boundaries between ranks are placed in
        the middle of the overlap of neighbouring ranks' needed ranges, i.e. the source partition follows the
        target-row partition and only a thin strip (the triangles straddling a row-block boundary) travels;
        cells nobody references are owned by nobody and never loaded."""
        needed_ids = np.asarray(needed_ids, np.int64)
        pr = cell_block(n_cells, world, rank)
        nlo = int(needed_ids[0]) if needed_ids.size else pr[0]
        nhi = int(needed_ids[-1] + 1) if needed_ids.size else pr[0]
        # banded numbering <=> the covering range of the needed ids is not much larger than their count
        empty = needed_ids.size == 0          # e.g. a row shard that lies entirely outside the mesh footprint
        want_range = empty or (nhi - nlo) <= range_slack * needed_ids.size
        votes = all_gather_object((bool(want_range), nlo, nhi, bool(empty)))
        mode = "range" if all(v[0] for v in votes) else "compact"
        if mode == "range" and ownership == "aligned":
            los, his = [v[1] for v in votes], [v[2] for v in votes]
            prev = min([v[1] for v in votes if not v[3]], default=0)
            for q in range(world):            # ranks that need nothing own an empty block right after their predecessor
                if votes[q][3]:
                    los[q] = his[q] = prev
                prev = his[q]
            if any(los[q] < los[q - 1] or his[q] < his[q - 1] for q in range(1, world)):
                ownership = "para_range"     # row blocks do not map to increasing id ranges
        if mode == "range" and ownership == "aligned":
            bnd = [los[0]]
            for q in range(1, world):
                b = (los[q] + his[q - 1]) // 2 if los[q] < his[q - 1] else los[q]
                bnd.append(max(b, bnd[-1]))
            bnd.append(max(his[-1], bnd[-1]))
            blocks = [(bnd[q], bnd[q + 1]) for q in range(world)]
        else:
            blocks = [cell_block(n_cells, world, q) for q in range(world)]
        c0, c1 = blocks[rank]
        if mode == "range":
            lo, hi = min(nlo, c0), max(nhi, c1)
            if not needed_ids.size:
                lo, hi = c0, c1
            spans = all_gather_object((lo, hi))
            s = HaloSchedule(rank, world, mode, hi - lo, (c0, c1), base=lo, own_pos=(c0 - lo, c1 - lo))
            for q in range(world):
                if q == rank:
                    s.send_ids.append((0, 0))
                    s.recv_pos.append((0, 0))
                    continue
                qlo, qhi = spans[q]
                a, b = max(qlo, c0), min(qhi, c1)          # what q wants from my block
                s.send_ids.append((a - c0, b - c0) if b > a else (0, 0))
                qc0, qc1 = blocks[q]
                a, b = max(lo, qc0), min(hi, qc1)          # what I want from q's block
                s.recv_pos.append((a - lo, b - lo) if b > a else (0, 0))
            return s
        lists = all_gather_object(needed_ids.astype(np.int32))
        s = HaloSchedule(rank, world, mode, int(needed_ids.size), (c0, c1))
        for q in range(world):
            theirs = lists[q]
            mine = theirs[(theirs >= c0) & (theirs < c1)]
            s.send_ids.append((mine - c0).astype(np.int32))
            qc0, qc1 = blocks[q]
            a = int(np.searchsorted(needed_ids, qc0))
            b = int(np.searchsorted(needed_ids, qc1))
            s.recv_pos.append((a, b))
        return s

    @staticmethod
    def build_owned(needed_ids, owned_ids, rank, world, all_gather_object):
        """The caller's own partition of the source cells (mpg_halo_build_owned): owned_ids = this rank's sorted unique global ids, any
        shape -- a coupled model's decomposition, or `owned_by_need` below.  Local space = the sorted needed ids (as in compact mode); a
        peer's cells arrive in id order and are scattered to their positions among them."""
        needed_ids, owned_ids = np.asarray(needed_ids, np.int32), np.asarray(owned_ids, np.int32)
        lists = all_gather_object((needed_ids, owned_ids))
        # every rank checks EVERY rank's needs (all lists are here), so that all refuse together -- a rank that went on alone would wait in
        # its first exchange for a peer that has given up (mpg_halo_build_owned does the same)
        all_owned = np.sort(np.concatenate([np.asarray(o, np.int32) for _, o in lists]), kind="stable")
        dup = np.nonzero(all_owned[1:] == all_owned[:-1])[0]
        if dup.size:      # mpg_halo_build_owned refuses a cell with two owners too: both transports accept the same partitions
            c = int(all_owned[dup[0]])
            raise ValueError("cell %d is owned by ranks %s: a partition gives every cell one owner" % (c, [q for q in range(world) if c in set(np.asarray(lists[q][1]).tolist())]))
        everyones = np.unique(all_owned)
        for q in range(world):
            theirs = np.asarray(lists[q][0], np.int32)
            orphan = np.setdiff1d(theirs, everyones, assume_unique=True)
            if orphan.size:
                raise ValueError("rank %d's rows reference cell %d, which no rank owns" % (q, int(orphan[0])))
        s = HaloSchedule(rank, world, "owned", int(needed_ids.size), (0, int(owned_ids.size)))
        s.owned_ids = owned_ids
        seen = np.zeros(needed_ids.size, bool)
        for q in range(world):
            their_needed, their_owned = lists[q]
            s.send_ids.append(np.searchsorted(owned_ids, np.intersect1d(their_needed, owned_ids, assume_unique=True)).astype(np.int32))
            pos = np.searchsorted(needed_ids, np.intersect1d(needed_ids, their_owned, assume_unique=True)).astype(np.int32)
            s.recv_pos.append(pos)
            seen[pos] = True
        if not seen.all():
            raise ValueError("rank %d's rows reference cell %d, which no rank owns" % (rank, int(needed_ids[~seen][0])))
        return s

    @staticmethod
    def owned_by_need(needed_ids, rank, world, all_gather_object):
        """A partition that follows the target rows whatever the cell numbering: every referenced cell belongs to the LOWEST rank whose rows
        reference it (cells nobody references belong to nobody).  What then travels is only what neighbouring row blocks share -- the
        `aligned` ownership of the range form, for meshes without banded numbering (a Morton-numbered global mesh in equal id blocks sends
        (N - 1) / N of every referenced value)."""
        needed_ids = np.asarray(needed_ids, np.int32)
        lists = all_gather_object(needed_ids)
        mine = needed_ids
        for q in range(rank):
            mine = np.setdiff1d(mine, lists[q], assume_unique=True)
        return mine.astype(np.int32)

    def counts(self):
        if self.mode == "range":
            send = [b - a for a, b in self.send_ids]
        else:
            send = [int(x.size) for x in self.send_ids]
        if self.mode == "owned":
            return send, [int(x.size) for x in self.recv_pos]
        recv = [b - a for a, b in self.recv_pos]
        return send, recv

    def plan(self, R, like):
        """Everything `exchange` needs for batches of R rows, allocated ONCE per (R, dtype, device): the packed send and
        receive buffers, their per-peer views, the split lists of the all-to-all-v and (gloo rehearsal on device tensors
        only) pinned host staging.  The per-step path then consists of copy launches and one collective -- no allocation,
        no list building, no torch.cat."""
        import torch
        # sources held cell-fast: rows [R][n] of scalars; in MPAS file order: [F][n][L] -- the exchanged element is one whole row of L
        # values (input_data.F90:630,645), so a neighbour's strip is one contiguous piece per field
        L = int(like.shape[2]) if like.dim() == 3 else 1
        key = (int(R), L if like.dim() == 3 else 0, like.dtype, like.device)
        p = self._plans.get(key)
        if p is not None:
            return p
        send_n, recv_n = self.counts()
        p = _HaloPlan()
        p.sendbuf = torch.empty(R * L * sum(send_n), dtype=like.dtype, device=like.device)
        p.recvbuf = torch.empty(R * L * sum(recv_n), dtype=like.dtype, device=like.device)
        p.in_splits, p.out_splits = [R * L * n for n in recv_n], [R * L * n for n in send_n]
        p.send_views, p.recv_views = [], []

        def shaped(buf, off, n):
            v = buf[off:off + R * L * n]
            return v.view(R, n, L) if like.dim() == 3 else v.view(R, n)
        off = 0
        for q in range(self.world):
            n = send_n[q]
            if n:
                view = shaped(p.sendbuf, off, n)
                if self.mode == "range":
                    p.send_views.append((view, self.send_ids[q], None))
                else:
                    ids = self.send_ids[q]
                    p.send_views.append((view, ids, torch.as_tensor(ids, dtype=torch.long, device=like.device)))
                off += R * L * n
        off = 0
        for q in range(self.world):
            n = recv_n[q]
            if n:
                where = self.recv_pos[q]
                if self.mode == "owned":      # positions among the needed ids, as an index tensor for the scatter
                    where = torch.as_tensor(where, dtype=torch.long, device=like.device)
                p.recv_views.append((shaped(p.recvbuf, off, n), where))
                off += R * L * n
        p.host_send = p.host_recv = None
        # what crosses the links: in the compact and owned forms the packed buffers also hold the rank's own share (a device copy)
        own_s = send_n[self.rank] if self.mode != "range" else 0
        own_r = recv_n[self.rank] if self.mode != "range" else 0
        p.bytes_sent = (sum(send_n) - own_s) * R * L * p.sendbuf.element_size()
        p.bytes_received = (sum(recv_n) - own_r) * R * L * p.recvbuf.element_size()
        self._plans[key] = p
        return p

    def exchange(self, own_rows, local_rows, pack_fn=None):
        """own_rows: tensor [R][n_own] of this rank's block (R = nfields*nlev rows) for sources held cell-fast, or
        [F][n_own][L] for sources in MPAS file order (any element type either way);
        local_rows: tensor [R][n_local] / [F][n_local][L], filled in place.  In range mode `own_rows` may be a view of
        `local_rows` (own data already in place).  pack_fn(own_rows, ids, out) gathers columns into `out` [R][len(ids)]([L])
        (HIP pack kernel on the GPU; index_select in the CPU tests when pack_fn is None)."""
        import torch
        import torch.distributed as dist
        p = self.plan(own_rows.shape[0], own_rows)
        for view, ids, ids_t in p.send_views:
            if ids_t is None:
                view.copy_(own_rows[:, ids[0]:ids[1]])
            elif pack_fn is not None:
                pack_fn(own_rows, ids, view)
            else:
                torch.index_select(own_rows, 1, ids_t, out=view)
        if self.world > 1:
            if p.sendbuf.is_cuda and dist.get_backend() == "gloo":
                # rehearsal transport (gloo moves host memory); the production transport is RCCL on device memory
                if p.host_send is None:
                    p.host_send = torch.empty(p.sendbuf.shape, dtype=p.sendbuf.dtype).pin_memory()
                    p.host_recv = torch.empty(p.recvbuf.shape, dtype=p.recvbuf.dtype).pin_memory()
                p.host_send.copy_(p.sendbuf)
                dist.all_to_all_single(p.host_recv, p.host_send, p.in_splits, p.out_splits)
                p.recvbuf.copy_(p.host_recv)
            else:
                dist.all_to_all_single(p.recvbuf, p.sendbuf, p.in_splits, p.out_splits)
        else:
            p.recvbuf.copy_(p.sendbuf)
        for view, where in p.recv_views:
            if self.mode == "owned":
                local_rows.index_copy_(1, where, view)
            else:
                local_rows[:, where[0]:where[1]].copy_(view)
        return local_rows


class _HaloPlan:
    __slots__ = ("sendbuf", "recvbuf", "in_splits", "out_splits", "send_views", "recv_views", "host_send", "host_recv",
                 "bytes_sent", "bytes_received")


class ShardedRegrid:
    """Rank-local piece of a row-sharded Mesh -> Grid regrid on this rank's GPU."""

    def __init__(self, mpas_mesh, target, regridmethod, rank, world, all_gather_object, ownership="aligned", transport="torch", id_file=None,
                 decomp_file=None):
        """ownership: "aligned" / "para_range" (the library's id-block partitions), "need" (every cell to the lowest rank whose rows reference it:
        a caller-defined partition, the owned halo form), "auto" (aligned when every rank's needed ids are banded, else need); decomp_file:
        an MPAS graph partition file (the namelist's block_decomp_file) -- the model's own decomposition, the owned form.
        transport "torch": schedule and exchange through torch.distributed (RCCL under the nccl backend, gloo in the CPU
        tests).  transport "cabi": the C-ABI's own verbs (mpg_comm_init / mpg_halo_build / mpg_halo_exchange_dev: librccl
        directly, what a C or Fortran host uses); the ranks meet through `id_file`."""
        from . import regrid as R
        self.rank, self.world = rank, world
        self.j0, self.j1 = row_block(target.ny, world, rank, nx=target.nx)
        self.grid = R.Grid.from_target(target, rows=(self.j0, self.j1))
        # with several ranks a rank's mesh holds only what its row block can see (mpg_mesh_create_window: same weights, geometry
        # ingest and Stores that shrink with the block; the reference gives every rank 1/N of the cells, model_grid.F90:423-438)
        self.mesh = R.Mesh.from_mpas(mpas_mesh, window_grid=self.grid if world > 1 else None)
        self.rh = R.regrid_store(self.mesh, self.grid, regridmethod)
        self.store_ms = self.rh.store_ms
        needed = self.rh.unique_sources()
        self.n_needed = int(needed.size)
        self._ids_dev = {}
        if ownership == "auto":     # banded numbering on every rank: the range form with aligned blocks; else the partition that follows the rows
            nlo, nhi = (int(needed[0]), int(needed[-1]) + 1) if needed.size else (0, 0)
            banded = all_gather_object(bool(needed.size == 0 or (nhi - nlo) <= 1.25 * needed.size))
            ownership = "aligned" if all(banded) else "need"
        owned = None
        if decomp_file is not None:     # the namelist's block_decomp_file: the model's own decomposition (model_grid.F90:423-438)
            ownership = "file"
            owned = read_block_decomp_file(decomp_file, mpas_mesh.nCells, world, rank)
        if ownership == "need":     # every cell to the lowest rank that references it: the caller-defined partition that follows the rows
            owned = HaloSchedule.owned_by_need(needed, rank, world, all_gather_object)
        if transport == "cabi":
            self.sched = CabiSchedule(self.rh, mpas_mesh.nCells, rank, world, id_file, ownership, owned_ids=owned)
            return
        if owned is not None:
            self.sched = HaloSchedule.build_owned(needed, owned, rank, world, all_gather_object)
            self.rh.localize()
            return
        self.sched = HaloSchedule.build(needed, mpas_mesh.nCells, rank, world, all_gather_object, ownership=ownership)
        if self.sched.mode == "range":
            self.rh.rebase(self.sched.base, self.sched.n_local)
        else:
            self.rh.localize()

    def _pack(self, own_rows, ids, out=None):
        """HIP gather of owned columns (mpg_pack_rows_dev) into `out` [R][len(ids)] / [F][len(ids)][L] (contiguous)."""
        import torch

        from . import comm as MC
        key = ids.ctypes.data
        if key not in self._ids_dev:
            self._ids_dev[key] = torch.as_tensor(ids, device=own_rows.device)
        ids_d = self._ids_dev[key]
        if out is None:
            out = torch.empty((own_rows.shape[0], ids.size) + tuple(own_rows.shape[2:]), dtype=own_rows.dtype, device=own_rows.device)
        return MC.pack_rows(own_rows, ids_d, out)

    def local_buffer(self, nfields, nlev, device, dtype=None, layout=None):
        """The rank's local source slab: [nfields * nlev][n_local] (cell-fast, input_data.F90:653-655) or [nfields][n_local][nlev]
        (MPAS file order, :630,645), float64 or float32."""
        import torch

        from . import regrid as R
        dtype = dtype or torch.float64
        if layout == R.LAYOUT_LEV_FAST:
            return torch.empty((nfields, self.sched.n_local, nlev), dtype=dtype, device=device)
        return torch.empty((nfields * nlev, self.sched.n_local), dtype=dtype, device=device)

    def own_view(self, local_rows):
        """Range mode: the slice of the local buffer that holds this rank's own cells (fill it in place)."""
        a, b = self.sched.own_pos
        return local_rows[:, a:b]

    def step(self, own_rows, local_rows, nlev, nfields, out=None):
        """Halo exchange + Regrid of one field batch.  Returns dst [nfields][nlev][rows][nx] (float64 from float64 sources, the
        sources' own type from float32 ones, as the Fortran driver's Regrid)."""
        import torch

        from . import regrid as R
        self.sched.exchange(own_rows, local_rows, pack_fn=self._pack if own_rows.is_cuda else None)
        layout = R.LAYOUT_LEV_FAST if local_rows.dim() == 3 else R.LAYOUT_CELL_FAST
        if local_rows.dtype == torch.float64:
            return self.rh.regrid(local_rows.view(-1), nlev=nlev, nfields=nfields, layout=layout, out=out)
        return self.rh.regrid_typed(local_rows.view(-1), nlev=nlev, nfields=nfields, layout=layout, out=out)

    def destroy(self):
        if isinstance(self.sched, CabiSchedule):
            self.sched.destroy()
        self.rh.release()
        self.mesh.destroy()
        self.grid.destroy()


class CabiSchedule:
    """The halo schedule and exchange of the C-ABI (csrc/mpg_comm.hip) behind the attributes ShardedRegrid / bench.py read
    from a HaloSchedule."""

    class _Bytes:
        bytes_sent = bytes_received = 0

    def __init__(self, rh, n_cells, rank, world, id_file, ownership, owned_ids=None):
        from . import comm as MC
        self.comm = MC.Comm(rank, world, id_file)
        self.halo = MC.Halo(self.comm, rh, n_cells, ownership, owned_ids=owned_ids)
        self.rank, self.world, self.mode = rank, world, self.halo.mode
        self.n_local, self.own, self.base, self.own_pos = self.halo.n_local, self.halo.own, self.halo.base, self.halo.own_pos
        self.owned_ids = owned_ids

    def exchange(self, own_rows, local_rows, pack_fn=None):
        return self.halo.exchange(own_rows, local_rows)

    def plan(self, R, like):
        b = CabiSchedule._Bytes()
        es = like.element_size() * (like.shape[2] if like.dim() == 3 else 1)     # file order: one whole row is the element
        b.bytes_sent = R * self.halo.sent_per_row * es
        b.bytes_received = R * self.halo.received_per_row * es
        return b

    def destroy(self):
        self.halo.destroy()
        self.comm.destroy()


class ShardedInterp:
    """The whole `interp_data` sequence (interp.F90:92-465) on a row shard of the target grid.

    Mirrors the reference's own data distribution for this step: every rank holds the full source fields (the
    reference reads every variable whole on every rank, input_data.F90:645) and owns a block of target rows; all
    Mesh -> Grid regrids are embarrassingly parallel.  The only neighbour dependence is the CENTER -> EDGE1/EDGE2
    destaggering of the rotated winds: a U/V point of a rank's first or last row needs the mass row just outside its
    block, so UMASS / VMASS exchange ONE halo row with each neighbour (P2P) and the stagger weights are built on the
    block extended by those rows.  Results are bit-identical to the single-GPU run.
    Ownership of staggered rows: rank r returns U rows [j0, j1) and V rows [j0, j1), the last rank also V row ny."""

    def __init__(self, mpas_mesh, target, rank, world):
        import copy

        from . import regrid as R
        self.rank, self.world, self.ny = rank, world, target.ny
        self.j0, self.j1 = row_block(target.ny, world, rank, nx=target.nx)
        self.e0, self.e1 = max(self.j0 - 1, 0), min(self.j1 + 1, target.ny)
        self.grid = R.Grid.from_target(target, rows=(self.j0, self.j1))
        self.mesh = R.Mesh.from_mpas(mpas_mesh, window_grid=self.grid if world > 1 else None)   # only what this row block can see
        self.grid_ext = R.Grid.from_target(target, rows=(self.e0, self.e1))
        self.target = copy.copy(target)               # rotation angles of the own rows
        self.cosa_ext = self.sina_ext = None
        if target.cosa is not None:
            self.target.cosa = np.ascontiguousarray(target.cosa[self.j0:self.j1])
            self.target.sina = np.ascontiguousarray(target.sina[self.j0:self.j1])
            self.cosa_ext, self.sina_ext = target.cosa[self.e0:self.e1], target.sina[self.e0:self.e1]   # ... and of the block with its halo rows

    def _halo_rows(self, mass):
        """mass [nz][rows][nx] of the own block -> [nz][e1-e0][nx] with the neighbours' boundary rows attached."""
        import torch
        import torch.distributed as dist
        lo, hi = self.j0 - self.e0, self.e1 - self.j1          # halo rows below / above (0 or 1)
        is_t = type(mass).__module__.startswith("torch")
        m = mass if is_t else torch.from_numpy(np.ascontiguousarray(mass))
        stage_cpu = m.is_cuda and self.world > 1 and dist.get_backend() == "gloo"
        ext = m.new_empty((m.shape[0], self.e1 - self.e0, m.shape[2]))
        ext[:, lo:lo + (self.j1 - self.j0)] = m
        if self.world > 1:
            ops, bufs = [], []

            def tr(x):
                return x.cpu() if stage_cpu else x
            if hi:   # my last row goes up, their first row comes down
                s, r = tr(m[:, -1].contiguous()), tr(m.new_empty((m.shape[0], m.shape[2])))
                ops += [dist.P2POp(dist.isend, s, self.rank + 1), dist.P2POp(dist.irecv, r, self.rank + 1)]
                bufs.append((r, ext.shape[1] - 1))
            if lo:
                s, r = tr(m[:, 0].contiguous()), tr(m.new_empty((m.shape[0], m.shape[2])))
                ops += [dist.P2POp(dist.isend, s, self.rank - 1), dist.P2POp(dist.irecv, r, self.rank - 1)]
                bufs.append((r, 0))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for r, row in bufs:
                ext[:, row] = r.to(ext.device)
        return ext if is_t else ext.numpy()

    def _destagger(self, grid, staggerloc, mass, nz):
        from . import regrid as R
        ext = self._halo_rows(mass)
        rh = R.regrid_store_grid(self.grid_ext, staggerloc)
        full = rh.regrid(ext.reshape(-1), nlev=nz)[0]              # [nz][ext rows (+1 for EDGE2)][nx (+1 for EDGE1)]
        rh.release()
        off = self.j0 - self.e0
        nrows = self.j1 - self.j0
        if staggerloc == R.STAGGERLOC_EDGE2 and self.j1 == self.ny:
            nrows += 1                                              # the last rank also owns the top V row
        return full[:, off:off + nrows]

    def _winds(self, um, vm, rot, nz):
        """interp.F90:291-328 in one pass on this rank's block: the halo rows that travel are those of the UNROTATED mass winds, the
        rotation (pointwise) and both destaggerings happen in mpg_wind_destagger_dev on the block extended by them -- the same bits as
        rotating first and exchanging rotated rows (the three-call route above).  None: not device-resident -> that route."""
        import torch

        from . import _lib as L, regrid as R
        ref = um if um is not None else vm
        if not (type(ref).__module__.startswith("torch") and ref.is_cuda and all(t is None or t.dtype == torch.float64 for t in (um, vm))):
            return None
        eu = self._halo_rows(um) if um is not None else None
        ev = self._halo_rows(vm) if vm is not None else None
        cosa = sina = None
        if rot:
            if getattr(self, "_rot_ext", None) is None or self._rot_ext[0].device != ref.device:
                self._rot_ext = (torch.as_tensor(np.ascontiguousarray(self.cosa_ext), device=ref.device),
                                 torch.as_tensor(np.ascontiguousarray(self.sina_ext), device=ref.device))
            cosa, sina = self._rot_ext
        rh_u = R.regrid_store_grid(self.grid_ext, R.STAGGERLOC_EDGE1) if eu is not None else None
        rh_v = R.regrid_store_grid(self.grid_ext, R.STAGGERLOC_EDGE2) if ev is not None else None
        try:
            u, v, _, _ = R.wind_destagger(rh_u, rh_v, cosa, sina, eu.contiguous() if eu is not None else None,
                                          ev.contiguous() if ev is not None else None, nz)
        except L.MpgError as e:
            if e.rc != L.MPG_ERR_UNSUPPORTED:
                raise
            return None
        finally:
            for rh in (rh_u, rh_v):
                if rh is not None:
                    rh.release()
        off, nrows = self.j0 - self.e0, self.j1 - self.j0
        return (u[:, off:off + nrows] if u is not None else None,
                v[:, off:off + nrows + (1 if self.j1 == self.ny else 0)] if v is not None else None)       # the last rank also owns the top V row

    def interp_data(self, inp, cfg):
        from . import interp as I
        return I.interp_data(self.mesh, self.grid, self.target, inp, cfg, destagger=self._destagger, winds=self._winds)

    def destroy(self):
        self.mesh.destroy()
        self.grid.destroy()
        self.grid_ext.destroy()
