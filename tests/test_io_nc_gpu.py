"""File-to-file run (mpassit.F90's main sequence): synthetic MPAS grid/history/diag files -> GPU regrid -> post-ops ->
WRF-style output file, read back with an independent NetCDF reader (scipy) and compared with the in-memory pipeline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_file_to_file_run_matches_the_in_memory_pipeline(gpu_lib, tmp_path):
    from scipy.io import netcdf_file

    from mpassit_amd import build, interp as I, io_nc, ncio, post, regrid as R, synth, target_grid as T, workloads
    build.build_ncio()
    m, _, nz, _ = workloads.workload("tiny")
    nsoil = 4
    rng = np.random.default_rng(17)
    f32 = lambda a: np.asarray(a, np.float32)           # noqa: E731  history files are single precision
    hist = {"xland": f32(np.floor(rng.uniform(1, 3, m.nCells))), "skintemp": f32(rng.uniform(250, 320, m.nCells)),
            "snow": f32(synth.snow_field(m.latCell, m.lonCell)[0]),
            "zgrid": f32(np.sort(rng.uniform(0, 2.0e4, (m.nCells, nz + 1)), axis=1)), "theta": f32(rng.uniform(280, 500, (m.nCells, nz))),
            "uReconstructZonal": f32(rng.normal(0, 10, (m.nCells, nz))), "uReconstructMeridional": f32(rng.normal(0, 10, (m.nCells, nz))),
            "pressure": f32(-np.sort(-rng.uniform(2.0e3, 1.0e5, (m.nCells, nz)), axis=1)), "rho": f32(rng.uniform(0.1, 1.2, (m.nCells, nz))),
            "tslb": f32(rng.uniform(260, 300, (m.nCells, nsoil)))}
    diag = {"t2m": f32(rng.uniform(250, 310, m.nCells)), "u10": f32(rng.normal(0, 8, m.nCells)), "v10": f32(rng.normal(0, 8, m.nCells)),
            "refl10cm": f32(rng.uniform(-30, 60, (m.nCells, nz)))}
    ter = rng.uniform(0, 3000, m.nCells)
    zs = [0.05, 0.25, 0.7, 1.5]
    gpath, hpath, dpath, opath = (tmp_path / n for n in ("grid.nc", "hist.nc", "diag.nc", "out.nc"))
    io_nc.write_mpas_files(gpath, hpath, m, ter, zs, hist, nz, nsoil, diag_path=dpath, diag=diag, xtime="2024-08-07_06:00:00")
    cfg = I.InterpConfig(interp_diag=True, wrf_mod_vars=True,
                         diag_list=[("t2m", "T2"), ("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM")],
                         hist_2d=[("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW")],
                         hist_3d=[("zgrid", "PHB"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"),
                                  ("pressure", "P_HYD"), ("rho", "MUB")], hist_soil=[("tslb", "TSLB")])
    nml = dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    target = T.define_target_grid_params("lambert", 181, 107, dy=30000.0, arrays=False, **nml)
    res = io_nc.run(gpath, hpath, opath, target, cfg, diag_path=dpath, namelist=nml, fmt=2)

    # the same job without files: float32 inputs widened, cell-fast layout, host-array target grid
    host = T.define_target_grid_params("lambert", 181, 107, dy=30000.0, **nml)
    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=nsoil, hgt=ter)
    inp.hist = {k: np.ascontiguousarray(v.astype(np.float64).T) for k, v in hist.items()}
    inp.diag = {k: np.ascontiguousarray(v.astype(np.float64).T) for k, v in diag.items()}
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(host)
    want = post.output_fields(I.interp_data(mesh, grid, host, inp, cfg), cfg)
    assert list(res) == list(want)

    f = netcdf_file(str(opath), "r", mmap=False)                      # independent reader of the classic format
    assert f.version_byte == 2 and f.dimensions["west_east"] == 180 and f.dimensions["south_north_stag"] == 107
    assert f.dimensions["bottom_top"] == nz and f.dimensions["Time"] is None and f.MAP_PROJ == 1
    assert abs(f.TRUELAT1 - 38.5) < 1e-6 and f.variables["XLAT"].units == b"degree_north"
    assert b"".join(f.variables["Times"][0]) == b"2024-08-07_06:00:00"
    np.testing.assert_allclose(f.variables["XLAT"][0], host.lat, rtol=0, atol=1e-5)
    np.testing.assert_allclose(f.variables["XLONG_U"][0], host.lon_u, rtol=0, atol=1e-5)
    np.testing.assert_allclose(f.variables["COSALPHA"][0], host.cosa, rtol=0, atol=1e-6)
    np.testing.assert_allclose(f.variables["ZS"][0], zs, rtol=1e-6)
    mf = f.variables["MAPFAC_M"][0]
    assert 0.9 < mf.min() and mf.max() < 1.2 and abs(mf[53, 90] - 1.0) < 1e-3          # ~1 at the true latitude
    for name, w in want.items():
        got = f.variables[name][0]
        if name == "P_TOP":
            assert got == w
            continue
        if name == "Z_C":
            assert got.shape == (nz + 1, host.ny, host.nx) and not got[nz].any()
            got = got[:nz]
        assert got.dtype.kind == "f" and got.dtype.itemsize == 4 and got.shape == w.shape      # NC_FLOAT (big-endian on disk)
        # device-generated vs host-generated grid coordinates differ in the last bits -> float32-level agreement
        np.testing.assert_allclose(got, w, rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(w).max())), err_msg=name)
    assert f.variables["U"].shape == (1, nz, host.ny, host.nx + 1) and f.variables["V"].shape == (1, nz, host.ny + 1, host.nx)
    f.close()

    # CDF-5 output is read back by this repo's reader
    res5 = io_nc.run(gpath, hpath, tmp_path / "out5.nc", target, cfg, diag_path=dpath, namelist=nml, fmt=5)
    with ncio.Reader(tmp_path / "out5.nc") as r:
        assert r.format == 5
        assert np.array_equal(r.get("T", rec=0), res5["T"]) and np.array_equal(r.get("TSLB", rec=0), res5["TSLB"])

    # device-side ingest / egress: variables cross as raw big-endian bytes and stay so on the GPU (the Regrid and the
    # post-ops read / write big-endian values, no swap pass), fields device resident in between -> the very same file,
    # byte for byte
    import torch
    resd = io_nc.run(gpath, hpath, tmp_path / "out5d.nc", target, cfg, diag_path=dpath, namelist=nml, fmt=5, device_io=True)
    assert list(resd) == list(res5)
    assert all(isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 for k, v in resd.items() if k != "P_TOP")
    assert (tmp_path / "out5d.nc").read_bytes() == (tmp_path / "out5.nc").read_bytes()
    inp_d, _, _ = io_nc.read_input_data(hpath, cfg, ter, diag_path=dpath, device="cuda")
    assert inp_d.hist["theta"].dtype == torch.float32 and tuple(inp_d.hist["theta"].shape) == (m.nCells, nz)
    assert inp_d.hist["theta"].mpg_be and inp_d.diag["u10"].mpg_be            # the file's bytes as they are stored
    assert np.array_equal(inp_d.hist["theta"].cpu().numpy().view(">f4"), hist["theta"])
    assert np.array_equal(inp_d.diag["u10"].cpu().numpy().view(">f4"), diag["u10"])
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("dtype", ["int16", "float32", "float64"])
def test_device_byte_swap_matches_numpy(gpu_lib, dtype):
    import torch

    from mpassit_amd import io_nc
    rng = np.random.default_rng(5)
    for n in (0, 1, 255, 65537 * 3 + 1):
        a = (rng.standard_normal(n) * 1000).astype(dtype)
        t = torch.as_tensor(a, device="cuda")
        io_nc.bswap_(t)
        torch.cuda.synchronize()
        assert t.cpu().numpy().tobytes() == a.byteswap().tobytes()
        io_nc.bswap_(t)
        assert np.array_equal(t.cpu().numpy(), a)


@pytest.mark.parametrize("fmt", [5, 4], ids=["cdf5", "netcdf4"])
def test_run_series_equals_one_run_per_file(gpu_lib, tmp_path, fmt):
    """io_nc.run_series (weights once, one captured time level replayed per file, next file prefetched by a reader
    thread) writes the same bytes as io_nc.run on every file of the series.  With NetCDF-4 files at both ends the reader thread and the
    writer are inside libhdf5 at the same time (the backend's lock) and the files hold the same variables as the classic ones."""
    from mpassit_amd import build, interp as I, io_nc, ncio, synth, target_grid as T, workloads
    build.build_ncio()
    if fmt == 4 and not ncio.has_netcdf4():
        pytest.skip("libmpassit_ncio was built without HDF5")
    m, _, nz, _ = workloads.workload("tiny")
    nsoil = 4
    f32 = lambda a: np.asarray(a, np.float32)           # noqa: E731
    ter = np.random.default_rng(1).uniform(0, 3000, m.nCells)
    cfg = I.InterpConfig(interp_diag=True, wrf_mod_vars=True, diag_list=[("t2m", "T2"), ("u10", "U10"), ("v10", "V10")],
                         hist_2d=[("xland", "XLAND"), ("snow", "SNOW"), ("skintemp", "TSK")],
                         hist_3d=[("zgrid", "PHB"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"),
                                  ("pressure", "P_HYD")], hist_soil=[("tslb", "TSLB")])
    nml = dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    target = T.define_target_grid_params("lambert", 181, 107, dy=30000.0, arrays=False, **nml)
    gpath = tmp_path / "grid.nc"
    jobs = []
    for k in range(3):
        rng = np.random.default_rng(100 + k)
        hist = {"xland": f32(np.floor(rng.uniform(1, 3, m.nCells))), "snow": f32(synth.snow_field(m.latCell, m.lonCell)[0] * (k + 1)),
                "skintemp": f32(rng.uniform(250, 320, m.nCells)), "zgrid": f32(np.sort(rng.uniform(0, 2.0e4, (m.nCells, nz + 1)), axis=1)),
                "theta": f32(rng.uniform(280, 500, (m.nCells, nz))), "uReconstructZonal": f32(rng.normal(0, 10, (m.nCells, nz))),
                "uReconstructMeridional": f32(rng.normal(0, 10, (m.nCells, nz))),
                "pressure": f32(-np.sort(-rng.uniform(2.0e3, 1.0e5, (m.nCells, nz)), axis=1)), "tslb": f32(rng.uniform(260, 300, (m.nCells, nsoil)))}
        diag = {"t2m": f32(rng.uniform(250, 310, m.nCells)), "u10": f32(rng.normal(0, 8, m.nCells)), "v10": f32(rng.normal(0, 8, m.nCells))}
        hp, dp = tmp_path / ("hist%d.nc" % k), tmp_path / ("diag%d.nc" % k)
        io_nc.write_mpas_files(gpath, hp, m, ter, [0.05, 0.25, 0.7, 1.5], hist, nz, nsoil, diag_path=dp, diag=diag,
                               xtime="2024-08-07_%02d:00:00" % k, fmt=fmt)
        jobs.append((hp, dp, tmp_path / ("series%d.nc" % k)))
    tm = {}
    assert io_nc.run_series(gpath, jobs, target, cfg, namelist=nml, timings=tm, fmt=fmt) == 3 and len(tm["files_s"]) == 3
    for k, (hp, dp, op) in enumerate(jobs):
        single = tmp_path / ("single%d.nc" % k)
        io_nc.run(gpath, hp, single, target, cfg, diag_path=dp, namelist=nml, device_io=True, fmt=fmt)
        if fmt != 4:
            assert op.read_bytes() == single.read_bytes(), "time level %d" % k
            continue
        with ncio.Reader(op) as a, ncio.Reader(single) as b:             # HDF5 containers are not byte-reproducible (times in object headers): by content
            assert a.format == b.format == 4 and list(a.vars) == list(b.vars) and a.dims == b.dims and list(a.atts()) == list(b.atts())
            for name in a.vars:
                assert a.get(name).tobytes() == b.get(name).tobytes(), (k, name)


def test_file_range_to_device_and_back(gpu_lib, tmp_path):
    """mpg_file_to_dev / mpg_dev_to_file: byte ranges at odd offsets and sizes (below one staging chunk, several chunks with
    a ragged tail, empty), both directions, plus the failure modes."""
    import ctypes as C

    import torch
    lib = gpu_lib.load()
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, (3 << 25) + 12345, dtype=np.uint8)          # ~100 MB + a ragged tail
    src = tmp_path / "blob.bin"
    src.write_bytes(data.tobytes())
    dst = tmp_path / "copy.bin"
    dst.write_bytes(b"\0" * data.size)
    for off, n in ((0, data.size), (7, 1), (13, 0), (4097, (1 << 25) + 1), (data.size - 1000, 1000), (3, 5 << 20)):
        t = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        gpu_lib.check(lib.mpg_file_to_dev(str(src).encode(), C.c_int64(off), C.c_int64(n), C.c_void_p(t.data_ptr()), None))
        got = t.cpu().numpy()
        assert np.array_equal(got[:n], data[off:off + n]) and not got[n:].any()
        gpu_lib.check(lib.mpg_dev_to_file(str(dst).encode(), C.c_int64(off), C.c_int64(n), C.c_void_p(t.data_ptr()), None))
    assert dst.read_bytes() == data.tobytes()                                # the first range covered everything
    t = torch.zeros(16, dtype=torch.uint8, device="cuda")
    assert lib.mpg_file_to_dev(str(tmp_path / "missing.bin").encode(), C.c_int64(0), C.c_int64(16), C.c_void_p(t.data_ptr()), None) == 2
    assert b"cannot open" in lib.mpg_last_error()
    assert lib.mpg_file_to_dev(str(src).encode(), C.c_int64(data.size - 8), C.c_int64(16), C.c_void_p(t.data_ptr()), None) == 2   # past the end
    assert b"short read" in lib.mpg_last_error()
    assert lib.mpg_dev_to_file(str(dst).encode(), C.c_int64(-1), C.c_int64(16), C.c_void_p(t.data_ptr()), None) == 2
