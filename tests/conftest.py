import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The Fortran driver reads an MPI / Slurm launcher's variables to find its rank (host_mod.F90 setup_ranks).  Should the test box itself have
# been started by such a launcher, the single-image runs of these tests must not take themselves for one rank of its job.
for _k in [k for k in os.environ if k in ("PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID", "SLURM_STEP_ID", "SLURM_NTASKS", "SLURM_PROCID", "SLURM_LOCALID") or
           k.startswith("OMPI_COMM_WORLD_")]:
    del os.environ[_k]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- shared synthetic cases (small enough for the oracle to finish in seconds) -------------------------
LAMBERT = dict(ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def global_mesh():
    from mpassit_amd import synth
    return synth.global_voronoi_mesh(20000)


@pytest.fixture(scope="session")
def conus_grid_30km():
    """181x107 namelist (180x106 mass points), 30 km Lambert over CONUS: sits inside the global mesh."""
    from mpassit_amd import target_grid as tg
    return tg.define_target_grid_params("lambert", 181, 107, dx=30000.0, dy=30000.0, **LAMBERT)


@pytest.fixture(scope="session")
def regional_case():
    """Regional hex mesh (rim + unmapped strip) with a target slightly LARGER than the mesh footprint."""
    from mpassit_amd import synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 151, 91, dx=30000.0, dy=30000.0, **LAMBERT)
    # mesh built for a smaller domain => outer target rows/cols fall outside the hull
    m = synth.regional_mesh_for_lambert(g.proj, 141, 81, 20000, margin=0.0)
    return m, g


@pytest.fixture(scope="session")
def gpu_lib():
    from mpassit_amd import _lib
    _lib.init(0)
    yield _lib
    _lib.finalize()


def mesh_xyz(o, m):
    lon_d, lat_d = o.mesh_coords_deg(m.lonCell, m.latCell)
    vlon_d, vlat_d = o.mesh_coords_deg(m.lonVertex, m.latVertex)
    return o.lonlat_deg_to_xyz(lon_d, lat_d), o.lonlat_deg_to_xyz(vlon_d, vlat_d)
