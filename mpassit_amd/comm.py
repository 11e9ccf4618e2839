"""ctypes face of the multi-GPU verbs of the C-ABI (include/mpassit_amd.h: mpg_comm_*, mpg_halo_*, mpg_gather_rows): RCCL
underneath, no torch.distributed.  `dist.py` is the torch.distributed form of the same schedule (bench.py, the gloo tests);
this module is what a C or Fortran host gets, driven from Python for the tests."""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import check


def plan_host(rank, needed_lists, n_cells, ownership="aligned"):
    """mpg_halo_plan_host: the halo schedule of `rank` as a pure function of every rank's sorted needed ids (no GPU)."""
    world = len(needed_lists)
    lists = [np.ascontiguousarray(x, np.int32) for x in needed_lists]
    n_needed = (C.c_int64 * world)(*[int(x.size) for x in lists])
    ptrs = (C.POINTER(C.c_int32) * world)(*[x.ctypes.data_as(C.POINTER(C.c_int32)) for x in lists])
    mode, n_local, base = C.c_int(), C.c_int64(), C.c_int64()
    own, own_pos = (C.c_int64 * 2)(), (C.c_int64 * 2)()
    arr = lambda: (C.c_int64 * world)()              # noqa: E731
    send_count, send_a, recv_a, recv_b = arr(), arr(), arr(), arr()
    cap = int(sum(x.size for x in lists)) + 1
    flat = (C.c_int32 * cap)()
    off = (C.c_int64 * (world + 1))()
    check(L.load().mpg_halo_plan_host(C.c_int(rank), C.c_int(world), C.c_int64(n_cells), C.c_int(0 if ownership == "aligned" else 1), n_needed,
                                      ptrs, C.byref(mode), C.byref(n_local), own, C.byref(base), own_pos, send_count, send_a, recv_a, recv_b, flat,
                                      C.c_int64(cap), off))
    res = dict(mode="range" if mode.value == 0 else "compact", n_local=n_local.value, own=(own[0], own[1]), base=base.value,
               own_pos=(own_pos[0], own_pos[1]), send_count=list(send_count), send_a=list(send_a), recv_pos=list(zip(recv_a, recv_b)))
    if mode.value == 1:
        f = np.frombuffer(flat, dtype=np.int32)
        res["send_ids"] = [f[off[q]:off[q + 1]].copy() for q in range(world)]
    return res


def plan_owned_host(rank, needed_lists, owned_lists):
    """mpg_halo_plan_owned_host: the owned-form schedule of `rank` as a pure function of every rank's sorted needed and owned ids
    (no GPU).  -> dict(n_local, send_ids [per peer: offsets into the rank's owned list], recv_ids [per peer: positions in its local space])."""
    world = len(needed_lists)
    nl = [np.ascontiguousarray(x, np.int32) for x in needed_lists]
    ol = [np.ascontiguousarray(x, np.int32) for x in owned_lists]
    P32 = C.POINTER(C.c_int32)
    n_needed = (C.c_int64 * world)(*[int(x.size) for x in nl])
    n_owned = (C.c_int64 * world)(*[int(x.size) for x in ol])
    np_ = (P32 * world)(*[x.ctypes.data_as(P32) for x in nl])
    op_ = (P32 * world)(*[x.ctypes.data_as(P32) for x in ol])
    cap_s, cap_r = int(ol[rank].size) * world + 1, int(nl[rank].size) + 1
    sflat, rflat = np.empty(cap_s, np.int32), np.empty(cap_r, np.int32)
    soff, roff = (C.c_int64 * (world + 1))(), (C.c_int64 * (world + 1))()
    n_local = C.c_int64()
    check(L.load().mpg_halo_plan_owned_host(C.c_int(rank), C.c_int(world), n_needed, np_, n_owned, op_, C.byref(n_local), sflat.ctypes.data_as(P32),
                                            C.c_int64(cap_s), soff, rflat.ctypes.data_as(P32), C.c_int64(cap_r), roff))
    return dict(n_local=n_local.value, send_ids=[sflat[soff[q]:soff[q + 1]].copy() for q in range(world)],
                recv_ids=[rflat[roff[q]:roff[q + 1]].copy() for q in range(world)])


class Comm:
    def __init__(self, rank=0, nranks=1, id_file=None, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            check(L.load().mpg_comm_init(C.c_int(rank), C.c_int(nranks), id_file.encode() if id_file else None, C.byref(self._h)))
        self.rank, self.nranks = rank, nranks

    def info(self):
        """mpg_comm_info -> (rank, nranks) as the communicator itself reports them."""
        r, n = C.c_int(), C.c_int()
        check(L.load().mpg_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def virtual(self, v_rank, v_nranks):
        """REHEARSAL on one GPU (mpg_comm_virtual): virtual rank v_rank of v_nranks on top of this one-rank communicator.  Drive every
        virtual rank from its own thread (`run_virtual_ranks`)."""
        h = C.c_void_p()
        check(L.load().mpg_comm_virtual(self._h, C.c_int(v_rank), C.c_int(v_nranks), C.byref(h)))
        return Comm(v_rank, v_nranks, _handle=h)

    def virtual_stats(self):
        """(groups, sends, receives, all-gathers) the virtual ranks of this communicator have really put through RCCL."""
        v = [C.c_int64() for _ in range(4)]
        check(L.load().mpg_comm_virtual_stats(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.empty((self.nranks,) + a.shape, a.dtype)
        check(L.load().mpg_comm_allgather(self._h, a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes), out.ctypes.data_as(C.c_void_p)))
        return out

    def gather_rows(self, rows, j0, j1, ny, root=0, stream=None):
        """rows: CUDA tensor [nlev][j1 - j0][nx] -> [nlev][ny][nx] on root (None elsewhere)."""
        import torch
        nlev, nx = rows.shape[0], rows.shape[-1]
        dst = torch.empty((nlev, ny, nx), dtype=rows.dtype, device=rows.device) if self.rank == root else None
        if stream is not None:
            check(L.load().mpg_gather_rows(self._h, C.c_void_p(rows.data_ptr()), C.c_int64(j0), C.c_int64(j1), C.c_int64(nx), C.c_int64(ny), C.c_int(nlev),
                                           C.c_int(rows.element_size()), C.c_void_p(dst.data_ptr() if dst is not None else 0), C.c_int(root), C.c_void_p(stream)))
            return dst
        check(L.load().mpg_gather_rows(self._h, C.c_void_p(rows.data_ptr()), C.c_int64(j0), C.c_int64(j1), C.c_int64(nx), C.c_int64(ny), C.c_int(nlev),
                                       C.c_int(rows.element_size()), C.c_void_p(dst.data_ptr() if dst is not None else 0), C.c_int(root),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return dst

    def destroy(self):
        if self._h:
            check(L.load().mpg_comm_destroy(self._h))
            self._h = C.c_void_p()


class Halo:
    """mpg_halo_build on a route handle (re-indexed in place to the local source space) + the exchange."""

    def __init__(self, comm, rh, n_cells, ownership="aligned", owned_ids=None):
        """owned_ids: this rank's sorted unique global cell ids -- the caller's own partition (mpg_halo_build_owned); else the library
        partitions by `ownership` ("aligned" | "para_range")."""
        self._h = C.c_void_p()
        if owned_ids is not None:
            self.owned_ids = np.ascontiguousarray(owned_ids, np.int32)
            check(L.load().mpg_halo_build_owned(comm._h, rh._h, C.c_int64(n_cells), self.owned_ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                C.c_int64(self.owned_ids.size), C.byref(self._h)))
        else:
            check(L.load().mpg_halo_build(comm._h, rh._h, C.c_int64(n_cells), C.c_int(0 if ownership == "aligned" else 1), C.byref(self._h)))
        rh._refresh()
        mode, n_local, base, sent, recv = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        own, own_pos = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        check(L.load().mpg_halo_info(self._h, C.byref(mode), C.byref(n_local), own, C.byref(base), own_pos, C.byref(sent), C.byref(recv)))
        self.mode, self.n_local, self.base = ("range", "compact", "owned")[mode.value], n_local.value, base.value
        self.own, self.own_pos, self.sent_per_row, self.received_per_row = (own[0], own[1]), (own_pos[0], own_pos[1]), sent.value, recv.value

    def exchange(self, own_rows, local_rows, stream=None):
        """own_rows: CUDA tensor [R][>= own block] (may be a view into local_rows in range form); local_rows [R][n_local] -- sources held
        cell-fast, R = fields x levels -- or [F][>= own block][L] and [F][n_local][L] for sources in file order: one whole row of L values
        is then the exchanged element."""
        import torch
        es = local_rows.element_size()
        if local_rows.dim() == 3:
            assert own_rows.dim() == 3 and own_rows.shape[2] == local_rows.shape[2] and own_rows.stride(2) == 1 and own_rows.stride(1) == own_rows.shape[2]
            es *= local_rows.shape[2]
            own_ld = own_rows.stride(0) // local_rows.shape[2]
        else:
            own_ld = own_rows.stride(0)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(L.load().mpg_halo_exchange_dev(self._h, C.c_void_p(own_rows.data_ptr()), C.c_int64(own_ld),
                                             C.c_void_p(local_rows.data_ptr()), C.c_int(local_rows.shape[0]), C.c_int(es), C.c_void_p(s)))
        return local_rows

    def destroy(self):
        if self._h:
            check(L.load().mpg_halo_destroy(self._h))
            self._h = C.c_void_p()


def pack_rows(own_rows, ids_dev, out, stream=None):
    """mpg_pack_rows_dev: out[k][i] = own_rows[k][ids[i]] -- rows of scalars ([R][n]) or of whole file-order rows ([F][n][L])."""
    import torch
    es = own_rows.element_size() * (own_rows.shape[2] if own_rows.dim() == 3 else 1)
    ld = own_rows.stride(0) // (own_rows.shape[2] if own_rows.dim() == 3 else 1)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    check(L.load().mpg_pack_rows_dev(C.c_void_p(own_rows.data_ptr()), C.c_int64(ld), C.c_int(own_rows.shape[0]), C.c_int(es), C.c_void_p(ids_dev.data_ptr()),
                                     C.c_int64(ids_dev.numel()), C.c_void_p(out.data_ptr()), C.c_void_p(s)))
    return out


def run_virtual_ranks(n, fn, timeout=120.0):
    """Run fn(v_rank) for v_rank = 0 .. n - 1, one host thread each (ctypes releases the GIL inside the library, so the threads
    meet in the library's rendezvous); returns the list of results, re-raises the first exception."""
    import threading
    res, err = [None] * n, [None] * n

    def body(r):
        try:
            res[r] = fn(r)
        except BaseException as e:     # noqa: BLE001 -- carried to the caller's thread
            err[r] = e
    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    if any(t.is_alive() for t in th):
        raise TimeoutError("virtual ranks did not finish within %.0f s" % timeout)
    for e in err:
        if e is not None:
            raise e
    return res
