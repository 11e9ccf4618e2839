"""The RCCL calls of the multi-GPU path, on the one GPU a test box has: a process group with backend "nccl" (= RCCL on
ROCm) and world size 1 still goes through RCCL's communicator set-up and collective entry points, so the exact calls
bench.py / dist.py make at N > 1 -- device-bound init, all_gather_object, all_to_all_single with split lists on float64
CUDA tensors issued on a side stream, barrier, MAX all_reduce of a device scalar -- are checked for this torch / RCCL
build.  (Data really crossing xGMI needs the driver's multi-GPU node.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    out = [None]
    dist.all_gather_object(out, {"needed": [1, 2, 3]})
    assert out[0] == {"needed": [1, 2, 3]}
    side = torch.cuda.Stream(device=dev)
    send = torch.arange(715 * 7, dtype=torch.float64, device=dev)
    recv = torch.empty_like(send)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_to_all_single(recv, send, [send.numel()], [send.numel()])
        ev = torch.cuda.Event(); ev.record(side)
    torch.cuda.current_stream().wait_event(ev)
    assert torch.equal(recv, send)
    empty = torch.empty(0, dtype=torch.float64, device=dev)
    dist.all_to_all_single(torch.empty_like(empty), empty, [0], [0])          # a rank with nothing to exchange
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    # the halo schedule itself, one rank: everything is owned locally, the exchange is a copy
    from mpassit_amd import _lib, dist as mdist, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, _ = workloads.workload("tiny")
    def ago(obj):
        o = [None]; dist.all_gather_object(o, obj); return o
    sr = mdist.ShardedRegrid(m, g, R.REGRIDMETHOD_BILINEAR, 0, 1, ago)
    assert sr.rh.n_dst == g.nx * g.ny
    sr.destroy()
    dist.destroy_process_group()
    print("rccl single-rank ok")
""") % ROOT


def test_rccl_entry_points_with_one_rank(gpu_lib):
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
