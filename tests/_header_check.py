"""Header parity of an output file against the list transcribed from the reference's writer (tests/golden/wrf_output_header.json)."""
import json
import os

from _cdf_header import parse

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrf_output_header.json")))


def check_output_header(path, interp_diag, proj_code, start_time, valid_time=None, config_dt=None, physics=None, prec_acc_dt=None):
    """dims, global attributes (names, stored types, ORDER), time variables and the coordinates attribute of every field of
    `path` against the golden list; returns the parsed header."""
    h = parse(path)
    assert [d for d, _ in h["dims"]] == GOLD["dims"]
    assert dict(h["dims"])["StrLen"] == GOLD["strlen"] and dict(h["dims"])["Time"] == 0          # Time = the record dimension
    want = [g for g in GOLD["global_attributes"] if len(g) == 2 or interp_diag]
    got = [(n, t) for n, t, _ in h["gatts"]]
    assert got == [(g[0], g[1]) for g in want], "global attributes differ from write_data.F90:196-308:\n%s\n%s" % (got, want)
    ga = {n: v for n, _, v in h["gatts"]}
    assert ga["MAP_PROJ_CHAR"] == GOLD["map_proj_char"][str(proj_code)] and ga["MAP_PROJ"] == [proj_code]
    assert ga["SIMULATION_START_DATE"] == start_time and ga["START_DATE"] == start_time
    assert ga["DX"] == ga["DY"] and ga["POL_ELAT"] == ga["POLE_LAT"] and ga["MOAD_CEN_LAT"] == ga["CEN_LAT"]
    assert ga["I_PARENT_START"] == [1] and ga["WEST-EAST_PATCH_END_STAG"] == [ga["WEST-EAST_PATCH_END_UNSTAG"][0] + 1]
    assert ga["WEST-EAST_GRID_DIMENSION"] == ga["WEST-EAST_PATCH_END_STAG"]
    if config_dt is not None:
        assert ga["DT"] == [config_dt]
    if physics is not None:
        assert (ga["SF_SURFACE_PHYSICS"], ga["MP_PHYSICS"], ga["CU_PHYSICS"]) == tuple([p] for p in physics)
    if interp_diag and prec_acc_dt is not None:
        assert ga["PREC_ACC_DT"] == [prec_acc_dt]
    vs = {v["name"]: v for v in h["vars"]}
    for name, spec in GOLD["time_variables"].items():
        v = vs[name]
        assert v["type"] == spec["type"] and v["dims"] == spec["dims"], (name, v)
        atts = {n: (val[0] if isinstance(val, list) else val) for n, _, val in v["atts"]}
        exp = {k: (x.replace("<start_time>", start_time) if isinstance(x, str) else x) for k, x in spec["atts"].items()}
        assert atts == exp, (name, atts, exp)
        assert [n for n, _, _ in v["atts"]] == list(spec["atts"]), "attribute order of %s" % name
    for name in GOLD["grid_variables"]:
        assert name in vs and vs[name]["type"] == "float"
    for v in h["vars"]:                               # every field points at coordinate variables that exist
        c = [val for n, _, val in v["atts"] if n == "coordinates"]
        if c and "XTIME" in c[0]:
            assert c[0] == GOLD["field_coordinates"]
            for cv in c[0].split():
                assert cv in vs, "%s: coordinates names %s, which is not in the file" % (v["name"], cv)
    return h
