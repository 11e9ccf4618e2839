"""Host logic mirrored from the reference: variable lists and the field -> method classification."""
import os

import pytest

from mpassit_amd import fields as F

# data fixtures: the reference's default lists (parm/diaglist, histlist_2d, histlist_3d, histlist_soil)
HIST_2D = [("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW"), ("snowh", "SNOWH"), ("sst", "SST")]
HIST_3D = [("zgrid", "PHB"), ("w", "W"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"),
           ("qv", "QVAPOR"), ("qc", "QCLOUD"), ("qr", "QRAIN"), ("qi", "QICE"), ("qs", "QSNOW"), ("qg", "QGRAUP"),
           ("ni", "QNICE"), ("nr", "QNRAIN"), ("pressure", "P_HYD"), ("rho", "MUB")]
SOIL = [("tslb", "TSLB"), ("smois", "SMOIS"), ("sh2o", "SH2O")]


def test_read_varlist(tmp_path):
    p = tmp_path / "histlist_2d"
    p.write_text("xland\t\t\tXLAND\n\nskintemp   TSK\nsnow\tSNOW\n   \n")
    assert F.read_varlist(str(p)) == [("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW")]
    (tmp_path / "empty").write_text("\n\n")
    assert F.read_varlist(str(tmp_path / "empty")) == []          # "ok if there are no fields" (input_data.F90:1181)
    with pytest.raises(FileNotFoundError):
        F.read_varlist(str(tmp_path / "missing"))
    (tmp_path / "bad").write_text("onlyonecolumn\n")
    with pytest.raises(ValueError):
        F.read_varlist(str(tmp_path / "bad"))


def test_default_lists_classification():
    h = F.classify_hist(HIST_2D, HIST_3D, SOIL, wrf_mod_vars=True)
    assert [n for n, _ in h.cons_2d] == ["snow", "snowh"]
    assert [n for n, _ in h.nstd_2d] == ["xland"]
    assert [n for n, _ in h.patch_2d] == ["skintemp", "sst"]
    assert [n for n, _ in h.nzp1_3d] == ["zgrid", "w"]
    assert h.do_u_interp and h.do_v_interp
    assert len(h.nz_3d) == 11 and "uReconstructZonal" not in [n for n, _ in h.nz_3d]
    assert F.soil_method(h) == "nearest"                             # default behaviour = nearest-neighbour soil (App. C3)
    h2 = F.classify_hist(HIST_2D, HIST_3D, SOIL, wrf_mod_vars=False)  # winds are ordinary nz fields (input_data.F90:898-903)
    assert not h2.do_u_interp and len(h2.nz_3d) == 13


def test_soil_method_fall_through():
    assert F.soil_method(F.classify_hist([("snow", "SNOW")], [], SOIL, False)) == "conserve"
    assert F.soil_method(F.classify_hist([("sst", "SST")], [], SOIL, False)) == "bilinear"
    assert F.soil_method(F.classify_hist([("snow", "SNOW"), ("landmask", "LANDMASK")], [], SOIL, False)) == "nearest"
    h = F.classify_hist([], [("vorticity", "VORT")], [], False)
    assert [n for n, _ in h.vert_3d] == ["vorticity"]


def test_diag_wind_indices():
    diag = [("rainc", "RAINC"), ("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM")]
    assert F.diag_wind_indices(diag) == (1, 2)
    assert F.diag_wind_indices(diag[:1]) == (None, None)
