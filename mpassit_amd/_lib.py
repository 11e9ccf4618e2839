"""ctypes binding of libmpassit_amd.so (the C-ABI of include/mpassit_amd.h).

There is no CPU implementation behind this module: if the HIP library is missing, or no GPU is
present at `init()`, every entry point raises.  Nothing here imports the test oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MPASSIT_AMD_LIB") or os.path.join(_HERE, "libmpassit_amd.so")   # override: A/B builds of experiments

# every symbol include/mpassit_amd.h declares (tests check they are all exported)
SYMBOLS = [
    "mpg_init", "mpg_finalize", "mpg_last_error", "mpg_device_info", "mpg_mesh_create", "mpg_mesh_create_window", "mpg_mesh_window_info", "mpg_mesh_destroy",
    "mpg_grid_create", "mpg_grid_attach_proj", "mpg_grid_destroy", "mpg_regrid_store", "mpg_regrid_store_grid", "mpg_regrid_store_begin", "mpg_regrid_store_grid_begin", "mpg_regrid",
    "mpg_regrid_dev", "mpg_regrid_typed_dev", "mpg_regrid_typed", "mpg_handle_release", "mpg_rotate_winds", "mpg_rotate_winds_dev", "mpg_wind_destagger_dev", "mpg_wind_destagger", "mpg_handle_info",
    "mpg_handle_from_weights", "mpg_handle_get_weights", "mpg_handle_get_csr", "mpg_mesh_get_triangles", "mpg_handle_unique_sources",
    "mpg_handle_localize", "mpg_handle_rebase", "mpg_pack_dev", "mpg_handle_store_ms", "mpg_regrid_bundle_typed_dev", "mpg_regrid_bundle_typed", "mpg_handle_store_path", "mpg_tune", "mpg_handle_pole_count", "mpg_handle_kernel_choice", "mpg_handle_tile_stats",
    "mpg_handle_get_pole", "mpg_bswap_dev", "mpg_file_to_dev", "mpg_dev_to_file", "mpg_dev_alloc", "mpg_dev_free", "mpg_dev_upload", "mpg_dev_download", "mpg_post_cast_dev", "mpg_post_layer_mean_dev", "mpg_post_ptop_dev", "mpg_post_ptop_parts_dev", "mpg_grid_create_proj", "mpg_grid_get_coords",
    "mpg_grid_get_rotang", "mpg_grid_get_mapfac", "mpg_grid_rotang_dev", "mpg_handle_source_range", "mpg_mesh_set_source_window",
    "mpg_comm_init", "mpg_comm_destroy", "mpg_comm_info", "mpg_comm_allgather", "mpg_halo_build", "mpg_halo_info", "mpg_halo_exchange_dev",
    "mpg_halo_destroy", "mpg_gather_rows", "mpg_halo_plan_host", "mpg_comm_idfile_verdict", "mpg_pack_rows_dev", "mpg_comm_virtual",
    "mpg_comm_virtual_stats", "mpg_handle_store_stats", "mpg_debug_scan_i32", "mpg_device_count", "mpg_warmup_wait", "mpg_halo_build_owned", "mpg_halo_plan_owned_host",
]

MPG_SUCCESS = 0
MPG_ERR_INVALID_ARG, MPG_ERR_UNSUPPORTED = 2, 4
REGRIDMETHOD_BILINEAR, REGRIDMETHOD_CONSERVE, REGRIDMETHOD_NEAREST_STOD = 0, 1, 2
MESHLOC_ELEMENT, MESHLOC_NODE = 0, 1
STAGGERLOC_CENTER, STAGGERLOC_EDGE1, STAGGERLOC_EDGE2, STAGGERLOC_CORNER = 0, 1, 2, 3
LAYOUT_CELL_FAST, LAYOUT_LEV_FAST = 0, 1
GRID_PERIODIC_I, GRID_NO_SOUTH_POLE, GRID_NO_NORTH_POLE = 1, 2, 4

_lib = None
_initialized = False


class MpgError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("libmpassit_amd rc=%d: %s" % (rc, msg))
        self.rc = rc


def load():
    """dlopen the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; when torch shares the process it
        # must be loaded FIRST so that this library binds to the same runtime (two runtimes in one process
        # leave the second one without a device).  C / Fortran callers have no torch and use /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "%s not found: build the HIP extension first (python -m mpassit_amd.build); "
                "mpassit_amd has no CPU fallback" % SO_PATH)
        L = C.CDLL(SO_PATH)
        L.mpg_last_error.restype = C.c_char_p
        L.mpg_comm_idfile_verdict.restype = C.c_char_p
        for name in SYMBOLS:
            if name not in ("mpg_last_error", "mpg_comm_idfile_verdict"):
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def check(rc):
    if rc != MPG_SUCCESS:
        raise MpgError(rc, load().mpg_last_error().decode("utf-8", "replace"))


def init(device=0):
    """ESMF_Initialize equivalent (mpassit.F90:84).  Requires a HIP device."""
    global _initialized
    check(load().mpg_init(C.c_int(device)))
    _initialized = True


def finalize():
    global _initialized
    if _lib is not None:
        check(_lib.mpg_finalize())
    _initialized = False


def device_info():
    buf = C.create_string_buffer(64)
    ncu, hbm = C.c_int(), C.c_int64()
    check(load().mpg_device_info(buf, C.c_int(64), C.byref(ncu), C.byref(hbm)))
    return buf.value.decode(), ncu.value, hbm.value


def tune(key, value):
    check(load().mpg_tune(key.encode(), C.c_int(int(value))))
