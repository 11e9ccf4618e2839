import numpy as np, torch, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from mpassit_amd import _lib, regrid as R, synth, target_grid as tg
from conftest import LAMBERT
_lib.init(0)
g = tg.define_target_grid_params("lambert", 151, 91, dx=30000.0, dy=30000.0, **LAMBERT)
m = synth.regional_mesh_for_lambert(g.proj, 141, 81, 20000, margin=0.0)
mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
rh_b = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
rh_n = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
rh_c = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
rh_u = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1)
nz = 5
dev = torch.device("cuda")
rng = np.random.default_rng(9)
s3 = torch.as_tensor(rng.standard_normal((nz, m.nCells)), device=dev)
s2 = torch.as_tensor(np.floor(rng.uniform(0, 20, m.nCells)), device=dev)
o3 = torch.empty((1, nz, g.ny, g.nx), dtype=torch.float64, device=dev)
oc = torch.empty((1, 1, g.ny, g.nx), dtype=torch.float64, device=dev)
on = torch.empty_like(oc)
ou = torch.empty((1, nz, g.ny, g.nx + 1), dtype=torch.float64, device=dev)
of = torch.empty((1, nz, g.ny, g.nx), dtype=torch.float32, device=dev)
steps = {"f64": lambda: rh_b.regrid(s3.view(-1), nlev=nz, out=o3),
         "u": lambda: rh_u.regrid(o3.view(-1), nlev=nz, out=ou),
         "n": lambda: rh_n.regrid(s2, nlev=1, out=on),
         "c": lambda: rh_c.regrid(s2, nlev=1, out=oc),
         "typed": lambda: rh_b.regrid_typed(s3.view(-1), nlev=nz, out_dtype=torch.float32, offset=-300.0, out=of)}
for name, fn in steps.items():
    fn(); torch.cuda.synchronize()
    print(name, "choice", rh_b.kernel_choice())
for name, fn in steps.items():
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                fn()
        print(name, "captured OK")
    except Exception as e:
        print(name, "FAILED", str(e)[:200])
        break
