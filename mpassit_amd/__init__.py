"""mpassit_amd -- MI355X-native replacement of MPASSIT's regrid hot path (interp.F90 + the ESMF regrid engine).

Product code only: HIP kernels + C-ABI (`csrc/`, `include/mpassit_amd.h`), Fortran host side (`fortran/`) and the
Python host mirror used by tests and bench (`regrid`, `interp`, `fields`, `target_grid`, `dist`).  The CPU oracle
lives outside the package (`oracle/`) and is never imported from here.  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
