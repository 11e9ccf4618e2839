"""Named synthetic workloads (BASELINE.json `configs`) shared by bench.py, smoke() and the tests."""
from . import synth
from . import target_grid as tg

README_LAMBERT = dict(dx=3000.0, dy=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)

# parm/histlist_3d with wrf_mod_vars=.false.: 13 nz-fields (incl. uReconstructZonal/Meridional), 2 nzp1
HISTLIST_3D_NZ = ["theta", "uReconstructZonal", "uReconstructMeridional", "qv", "qc", "qr", "qi", "qs", "qg", "ni", "nr",
                  "pressure", "rho"]


# The reference's default variable lists (parm/histlist_2d, histlist_3d, histlist_soil, diaglist: MPAS name -> output name)
# as a whole-job workload: tools/run_config.py and bench.py's `job` object run interp_data over them
JOB_HIST_2D = [("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW"), ("snowh", "SNOWH"), ("sst", "SST")]
JOB_HIST_3D = [("zgrid", "PHB"), ("w", "W"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"), ("qv", "QVAPOR"),
               ("qc", "QCLOUD"), ("qr", "QRAIN"), ("qi", "QICE"), ("qs", "QSNOW"), ("qg", "QGRAUP"), ("ni", "QNICE"), ("nr", "QNRAIN"),
               ("pressure", "P_HYD"), ("rho", "MUB")]
JOB_SOIL = [("tslb", "TSLB"), ("smois", "SMOIS"), ("sh2o", "SH2O")]
JOB_DIAG = [("rainc", "RAINC"), ("rainnc", "RAINNC"), ("snowncv", "SNOWNCV"), ("rainncv", "RAINNCV"), ("graupelncv", "GRAUPELNCV"),
            ("prec_acc_c", "PREC_ACC_C"), ("prec_acc_nc", "PREC_ACC_NC"), ("snow_acc_nc", "SNOW_ACC_NC"), ("refl10cm", "REFL_10CM"),
            ("refl10cm_max", "COMPOSITE_REFL_10CM"), ("refl10cm_1km", "REFL_10CM_1KM"), ("refl10cm_1km_max", "REFL_10CM_1KM_MAX"),
            ("u10", "U10"), ("v10", "V10"), ("q2", "Q2"), ("t2m", "T2"), ("th2m", "TH2"), ("updraft_helicity_max", "UP_HELI_MAX"),
            ("w_velocity_max", "W_UP_MAX"), ("surface_pressure", "PSFC")]


def conus_lambert_grid(nx=1801, ny=1061, **over):
    """README.md:53-73 namelist: 1801x1061 (staggered) 3-km Lambert grid -> 1800x1060 mass points."""
    p = dict(README_LAMBERT)
    p.update(over)
    return tg.define_target_grid_params("lambert", nx, ny, **p)


def workload(name, arrays=True):
    """-> (MpasMesh, TargetGrid, nlev, description).  arrays=False: the TargetGrid carries only the projection
    (coordinates are then generated on the device, regrid.Grid.from_proj); regional Lambert workloads only."""
    if not arrays:
        m, g, nlev, desc = workload(name)
        if g.proj.code != tg.PROJ_LC or name == "tiny":
            raise KeyError("arrays=False is wired for the CONUS Lambert workloads")
        return m, conus_lambert_grid(arrays=False), nlev, desc
    if name == "c4_3m_regional":
        g = conus_lambert_grid()
        m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000)
        return m, g, 55, "3.0 M-cell regional hex mesh x 55 levels -> 1801x1061 Lambert (1800x1060 mass points)"
    if name == "x_c4_nx1793":
        # extra (alignment experiment): configuration 4's mesh under a grid 1792 mass points wide -- every row of the output
        # planes starts on a 128-byte line for float32 and float64 alike
        g = conus_lambert_grid(nx=1793)
        m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000)
        return m, g, 55, "3.0 M-cell regional hex mesh x 55 levels -> 1793x1061 Lambert (1792x1060 mass points)"
    if name == "x_c4_1799x1059":
        # extra (round 6, plane alignment): configuration 4's mesh under HRRR's own grid size -- 1799 x 1059 mass points, an odd
        # number of points per level: plane k of every result starts k * 84 (float32) / k * 40 (float64) bytes (mod 128) into a line
        g = conus_lambert_grid(nx=1800, ny=1060)
        m = synth.regional_mesh_for_lambert(conus_lambert_grid().proj, 1801, 1061, 3_000_000)
        return m, g, 55, "3.0 M-cell regional hex mesh x 55 levels -> 1800x1060 Lambert (1799x1059 mass points: planes of an odd number of points)"
    if name.startswith("x_c4_rows"):
        # extra (strong-scaling rehearsal on one GPU): one rank's share of configuration 4 at N ranks = a block of
        # 1060 / N grid rows of the same 1800-wide grid over the same mesh, e.g. x_c4_rows133 for N = 8
        rows = int(name[len("x_c4_rows"):])
        g = conus_lambert_grid(ny=rows + 1)
        m = synth.regional_mesh_for_lambert(conus_lambert_grid().proj, 1801, 1061, 3_000_000)
        return m, g, 55, "3.0 M-cell regional hex mesh x 55 levels -> a %d-row block of the 1800-wide Lambert grid" % rows
    if name in ("x_c4_polar", "x_c4_mercator"):
        # extra (round 5: index-space Stores on the other two projections of the namelist, program_setup.F90:174-182): configuration
        # 4's sizes -- a 3.0 M-cell regional hex mesh under an 1801x1061 3-km grid -- on a polar stereographic grid with the north pole
        # inside it and on a Mercator grid across the date line
        if name == "x_c4_polar":
            g = tg.define_target_grid_params("polar", 1801, 1061, dx=3000.0, dy=3000.0, ref_lat=89.0, ref_lon=25.0, truelat1=75.0, stand_lon=-100.0)
        else:
            g = tg.define_target_grid_params("mercator", 1801, 1061, dx=3000.0, dy=3000.0, ref_lat=-8.0, ref_lon=179.0, truelat1=-15.0, stand_lon=179.0)
        m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000)
        return m, g, 55, "3.0 M-cell regional hex mesh x 55 levels -> 1801x1061 %s grid (1800x1060 mass points)" % name[5:]
    if name == "c2_655k_global":
        # BASELINE configs 2 and 3: the GLOBAL quasi-uniform 655 362-cell mesh (10*4^8 + 2 cells = MPAS x1.655362, SURVEY
        # s8(d)) under the README Lambert domain, which touches only 2-3 % of its cells; Morton-numbered.
        g = conus_lambert_grid()
        m = synth.icosahedral_mesh(8)
        return m, g, 55, "655 362-cell global icosahedral mesh (x1.655362) x 55 levels -> 1801x1061 Lambert (1800x1060 mass points)"
    if name == "x_655k_lattice":
        # extra workload (not a BASELINE config): 655 362 cells of a row-numbered regional lattice laid over the Lambert
        # domain -- every cell referenced, 2.9 target points per cell
        g = conus_lambert_grid()
        m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 655_362)
        return m, g, 55, "655 362-cell regional hex lattice x 55 levels -> 1801x1061 Lambert"
    if name == "c1_65k_global":
        g = tg.define_target_grid_params("lat-lon", 201, 201, dx=0.1, dy=0.1, ref_lat=30.0, ref_lon=-110.0, ref_x=1.0, ref_y=1.0,
                                         stand_lon=-110.0)
        m = synth.global_voronoi_mesh(65_536)
        return m, g, 10, "65 536-cell global Voronoi mesh x 10 levels -> 200x200 0.1-degree lat-lon"
    if name in ("c5_global_latlon", "c5_small"):
        # BASELINE config 5: global mesh -> global lat-lon grid (is_regional=.false.: 360/nx x 180/ny degrees, periodic in i,
        # poles closed, program_setup.F90:197-217 / model_grid.F90:684-696)
        # "3 M-cell mesh": the class-I geodesic grid of frequency 548 = 3 003 042 cells (round 6; until round 5 the bisection mesh of
        # 2 621 442 cells stood in: the bisection family has nothing between 2.6 M and 10.5 M)
        nx, ny = (3601, 1801) if name == "c5_global_latlon" else (361, 181)
        g = tg.define_target_grid_params("lat-lon", nx, ny, stand_lon=0.0, is_regional=False)
        m = synth.geodesic_mesh(548) if name == "c5_global_latlon" else synth.icosahedral_mesh(6)
        return m, g, 55, "%d-cell global %s mesh x 55 levels -> %dx%d global lat-lon (%dx%d mass points)" % (
            m.nCells, "geodesic (frequency 548)" if name == "c5_global_latlon" else "icosahedral", nx, ny, nx - 1, ny - 1)
    if name == "c5_2p6m_bisection":      # rounds 1-5's stand-in for configuration 5 (kept for same-mesh comparisons with their records)
        g = tg.define_target_grid_params("lat-lon", 3601, 1801, stand_lon=0.0, is_regional=False)
        m = synth.icosahedral_mesh(9)
        return m, g, 55, "%d-cell global icosahedral mesh x 55 levels -> 3601x1801 global lat-lon (3600x1800 mass points)" % m.nCells
    if name == "c4_3m_cutout":
        # configuration 4 numbered as a limited-area CUT-OUT of a global mesh: the regional cells in the order of a parent numbered along a
        # space-filling curve over the whole sphere (synth.cutout_cells) -- how a regional MPAS mesh made by MPAS-Limited-Area is numbered
        g = conus_lambert_grid()
        m = synth.cutout_cells(synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000))
        return m, g, 55, "3.0 M-cell regional hex mesh numbered as a cut-out of a Morton-numbered global mesh, x 55 levels -> 1801x1061 Lambert" 
    if name == "x_655k_global005":
        # extra workload (not a BASELINE config): a COARSE mesh under a FINE global grid -- 655 362 cells (30 km) -> 7200 x 3600 points (0.05 degrees):
        # 40 target points per cell, 94 % of the algorithmic bytes are stores.  The shape that showed the staged level-fast kernel's fixed row
        # slots (profiles/r05_lfu_npf.md); use --fields 4 (thirteen 55-level float64 fields of 25.9 M points are 148 GB)
        g = tg.define_target_grid_params("lat-lon", 7201, 3601, stand_lon=0.0, is_regional=False)
        m = synth.icosahedral_mesh(8)
        return m, g, 55, "655 362-cell global icosahedral mesh x 55 levels -> 7201x3601 global lat-lon (7200x3600 mass points)"
    if name == "c4_3m_morton":
        # configuration 4 with a REALISTIC cell numbering: the same 3.0 M cells, renumbered along a Morton curve
        # (locality-preserving, not row-banded -- what a production mesh reordered by a space-filling curve or a graph
        # partitioner looks like); c4_3m_regional numbers the lattice row by row, the best case for a cell-fast gather
        g = conus_lambert_grid()
        m = synth.morton_cells(synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000))
        return m, g, 55, "3.0 M-cell regional hex mesh, Morton-numbered cells, x 55 levels -> 1801x1061 Lambert (1800x1060 mass points)"
    if name.startswith("c4_3m_shuffled"):  # same mesh, cells renumbered at random (optionally in blocks: c4_3m_shuffled_b64)
        g = conus_lambert_grid()
        m = synth.regional_mesh_for_lambert(g.proj, 1801, 1061, 3_000_000)
        block = int(name.split("_b")[1]) if "_b" in name else 1
        return synth.shuffle_cells(m, block=block), g, 55, "3.0 M-cell regional mesh, cells renumbered at random (blocks of %d)" % block
    if name == "tiny":
        g = tg.define_target_grid_params("lambert", 181, 107, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                         truelat2=38.5, stand_lon=-97.5)
        m = synth.regional_mesh_for_lambert(g.proj, 181, 107, 30_000)
        return m, g, 8, "30 k-cell regional hex mesh x 8 levels -> 181x107 30-km Lambert"
    raise KeyError(name)
