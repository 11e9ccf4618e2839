"""The output file's header -- dimensions, global attributes (names, stored types, order), XTIME / ITIMESTEP / Times --
against the list transcribed from the reference's writer (write_data.F90:177-560, tests/golden/wrf_output_header.json).
CPU only: the Python writer with a stand-in for the device grid; the Fortran writer is held to the same list in
tests/test_fortran_driver_nc_gpu.py."""
import numpy as np
import pytest

from _header_check import check_output_header

NL = dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)


class _HostGrid:
    """What write_target_data asks of regrid.Grid (coords / mapfac / rotang), served from the host mirror of the projection."""

    def __init__(self, g):
        self.g = g

    def coords(self, st):
        from mpassit_amd import regrid as R
        g = self.g
        return {R.STAGGERLOC_CENTER: (g.lon, g.lat), R.STAGGERLOC_EDGE1: (g.lon_u, g.lat_u), R.STAGGERLOC_EDGE2: (g.lon_v, g.lat_v)}[st]

    def mapfac(self, st):
        return np.ones_like(self.coords(st)[0])

    def rotang(self):
        return self.g.cosa, self.g.sina


@pytest.mark.parametrize("kind,interp_diag", [("lambert", True), ("lambert", False), ("lat-lon", True)])
def test_python_writer_header(tmp_path, kind, interp_diag):
    from mpassit_amd import interp as I, io_nc, ncio, target_grid as tg
    if kind == "lambert":
        g = tg.define_target_grid_params("lambert", 21, 17, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                         stand_lon=-97.5)
    else:
        g = tg.define_target_grid_params("lat-lon", 21, 17, dx=0.5, dy=0.5, ref_lat=30.0, ref_lon=-110.0, ref_x=1.0, ref_y=1.0, stand_lon=-110.0)
    cfg = I.InterpConfig(interp_diag=interp_diag)
    res = {"T2": np.zeros((g.ny, g.nx), np.float32), "T": np.zeros((3, g.ny, g.nx), np.float32), "U": np.zeros((3, g.ny, g.nx + 1), np.float32)}
    meta = dict(start_time="2024-08-07_00:00:00", config_dt=20.0, lsm_scheme=3, mp_scheme=18, conv_scheme=16, diag_out_interval=1800)
    path = str(tmp_path / "out.nc")
    io_nc.write_target_data(path, g, _HostGrid(g), res, cfg, 3, 4, 4, valid_time="2024-08-07_06:00:00", namelist=NL, meta=meta)
    h = check_output_header(path, interp_diag, g.proj.code, "2024-08-07_00:00:00", config_dt=20.0, physics=(3, 18, 16), prec_acc_dt=1800)
    names = [v["name"] for v in h["vars"]]
    assert ("SINALPHA" in names) == (kind == "lambert")
    with ncio.Reader(path) as r:      # the reference subtracts start - valid (write_data.F90:1225): six hours in -> -360 minutes
        assert float(r.get("XTIME", rec=0)) == np.float32(-360.0) and int(r.get("ITIMESTEP", rec=0)) == -1080
    # no time step in the input files -> ITIMESTEP 0 (:1238), start time falls back to the valid time
    io_nc.write_target_data(path, g, _HostGrid(g), res, cfg, 3, 4, 4, valid_time="2024-08-07_06:00:00", namelist=NL, meta={})
    check_output_header(path, interp_diag, g.proj.code, "2024-08-07_06:00:00", config_dt=0.0, physics=(0, 0, 0), prec_acc_dt=0)
    with ncio.Reader(path) as r:
        assert float(r.get("XTIME", rec=0)) == 0.0 and int(r.get("ITIMESTEP", rec=0)) == 0


def test_meta_from_mpas_files(tmp_path):
    """read_file_meta: scheme names -> WRF option numbers exactly as input_data.F90:347-383 maps them; missing -> 0."""
    from mpassit_amd import io_nc, ncio
    p = str(tmp_path / "h.nc")
    with ncio.Writer(p, format=2) as w:
        w.def_dim("n", 1)
        w.put_att("config_start_time", "2024-01-01_00:00:00")
        w.put_att("config_dt", 12.5)
        w.put_att("config_lsm_scheme", "ruc")
        w.put_att("config_microp_scheme", "mp_nssl2m")
        w.put_att("config_convection_scheme", "cu_kain_fritsch")
        w.put_att("output_interval", np.int32(900))
        w.def_var("x", ncio.FLOAT, ("n",))
        w.put("x", np.zeros(1, np.float32))
    with ncio.Reader(p) as r:
        m = io_nc.read_file_meta(r, {})
        assert m == dict(start_time="2024-01-01_00:00:00", config_dt=12.5, lsm_scheme=3, mp_scheme=18, conv_scheme=1)
        assert io_nc.read_file_meta(r, {}, diag=True)["diag_out_interval"] == 900
    p2 = str(tmp_path / "bare.nc")
    with ncio.Writer(p2, format=2) as w:
        w.def_dim("n", 1)
        w.def_var("x", ncio.FLOAT, ("n",))
        w.put("x", np.zeros(1, np.float32))
    with ncio.Reader(p2) as r:
        assert io_nc.read_file_meta(r, {}) == dict(config_dt=0.0, lsm_scheme=0, mp_scheme=0, conv_scheme=0)
    assert io_nc.xtime_minutes("2024-08-07_00:00:00", "2024-08-08_01:30:00") == -1530.0
