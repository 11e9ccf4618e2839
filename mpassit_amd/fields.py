"""Variable lists and the field -> regrid-method classification of the reference.

Mirrors `read_varlist` (input_data.F90:1146-1194: two whitespace-separated columns, blank lines skipped)
and the hard-coded classification of `init_input_hist_fields` (input_data.F90:840-843,858-866,896-911):
  2-D:  {snow, snowh} -> conservative;  {ivgtyp, isltyp, xland, landmask} -> nearest;  rest -> bilinear
  3-D:  {zgrid, w} -> nz+1 levels;  {vorticity} -> node-located;  uReconstructZonal/Meridional -> staggered
        U/V only if wrf_mod_vars (:898-903);  rest -> nz levels
This decides which HIP kernel family serves each field ("classification rule is in scope").
"""
import os
from dataclasses import dataclass, field

CONS_VARS = ("snow", "snowh")                           # input_data.F90:840
NSTD_VARS = ("ivgtyp", "isltyp", "xland", "landmask")   # :841
NZP1_VARS = ("zgrid", "w")                              # :842
VERT_VARS = ("vorticity",)                              # :843


def read_varlist(path):
    """-> [(mpas_name, target_name), ...]; a missing file is an error like in the reference (:1160-1163)."""
    if not os.path.exists(path):
        raise FileNotFoundError("VARLIST FILE %s not exist" % path)
    out = []
    with open(path) as f:
        for line in f:
            if line.strip() == "":
                continue
            parts = line.split()
            if len(parts) < 2:
                raise ValueError("READING VARLIST FILE: need two columns in %r" % line)
            out.append((parts[0], parts[1]))
    return out


@dataclass
class HistFields:
    patch_2d: list = field(default_factory=list)   # bilinear 2-D
    cons_2d: list = field(default_factory=list)    # conservative 2-D
    nstd_2d: list = field(default_factory=list)    # nearest 2-D
    nz_3d: list = field(default_factory=list)      # bilinear, nz levels
    nzp1_3d: list = field(default_factory=list)    # bilinear, nz+1 levels
    vert_3d: list = field(default_factory=list)    # node-located bilinear (vorticity)
    soil: list = field(default_factory=list)       # nsoil levels, method by fall-through (interp.F90:436-447)
    do_u_interp: bool = False
    do_v_interp: bool = False


def classify_hist(list_2d, list_3d, list_soil, wrf_mod_vars):
    """Lists of (mpas_name, target_name) -> HistFields, preserving list order inside every class."""
    h = HistFields(soil=list(list_soil))
    for name, tgt in list_2d:
        if name in CONS_VARS:
            h.cons_2d.append((name, tgt))
        elif name in NSTD_VARS:
            h.nstd_2d.append((name, tgt))
        else:
            h.patch_2d.append((name, tgt))
    for name, tgt in list_3d:
        if wrf_mod_vars and name == "uReconstructZonal":
            h.do_u_interp = True
        elif wrf_mod_vars and name == "uReconstructMeridional":
            h.do_v_interp = True
        elif name in NZP1_VARS:
            h.nzp1_3d.append((name, tgt))
        elif name in VERT_VARS:
            h.vert_3d.append((name, tgt))
        else:
            h.nz_3d.append((name, tgt))
    return h


def soil_method(h):
    """Method the soil bundle inherits (interp.F90:436-441 uses whatever `method` holds)."""
    if h.nstd_2d:
        return "nearest"
    if h.cons_2d:
        return "conserve"
    return "bilinear"


def diag_wind_indices(diag_list):
    """u10/v10 positions in the diag bundle (input_data.F90:173-180); None when absent."""
    names = [n for n, _ in diag_list]
    return (names.index("u10") if "u10" in names else None, names.index("v10") if "v10" in names else None)
