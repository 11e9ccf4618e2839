"""Host-side mirror of the data section of `write_target_data` (write_data.F90:996-1498), without the NetCDF calls.

The reference gathers every regridded float64 field to rank 0, applies a few WRF-specific post-ops and hands the
result to nf90_put_var, which converts to the variables' NF90_FLOAT type.  Here the same arithmetic runs as device
epilogues (csrc/k_post.hip) on the device-resident fields and the result is the float32 array the file would hold:

  every field           float64 -> float32                                  write_data.F90:1008-1330
  T (wrf_mod_vars)      T - 300   (the `continue` at :1342 is a no-op)       :1339-1347
  MUB (wrf_mod_vars)    + MU  = 0                                            :1354-1360
  P_HYD (wrf_mod_vars)  + P_TOP (scalar), + PB = P_HYD                       :1362-1379
  PHB                   + Z_C(k) = (PHB(k+1)+PHB(k))/2, PHB = PHB*9.81       :1406-1424
  PHB (wrf_mod_vars)    + PH  = 0                                            :1427-1432
  (wrf_mod_vars)        + P   = 0                                            :1466-1475
Order of the returned dict = order of the writes.  NetCDF itself is out of scope (no library in the image).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import fields as F
from ._lib import check


def _dev(x):
    import torch
    if type(x).__module__.startswith("torch"):
        if not x.is_cuda:
            raise ValueError("post-ops run on the GPU: pass CUDA tensors or numpy arrays")
        return x.contiguous(), True
    return torch.as_tensor(np.ascontiguousarray(x, np.float64), device="cuda"), False


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def cast_f32(x, scale=1.0, offset=0.0, be=False):
    """(float)(x*scale + offset) -- nf90_put_var's float64 -> NF90_FLOAT conversion with the writer's affine post-op.
    be: store the float32 values big-endian, as the NetCDF classic output file holds them (CUDA tensors only; the
    result carries the attribute mpg_be = True)."""
    import torch
    d, was_t = _dev(x)
    if be and not was_t:
        raise ValueError("big-endian results are for the device flow: pass a CUDA tensor")
    out = torch.empty(d.shape, dtype=torch.float32, device=d.device)
    check(L.load().mpg_post_cast_dev(C.c_void_p(d.data_ptr()), C.c_int64(d.numel()), C.c_double(scale), C.c_double(offset),
                                     C.c_void_p(out.data_ptr()), C.c_int(int(be)), _stream()))
    if be:
        out.mpg_be = True
    return out if was_t else out.cpu().numpy()


def layer_mean_f32(x, be=False):
    """Z_C: [nlevp1][...] -> [nlevp1-1][...], 0.5*(x[k+1] + x[k]) (write_data.F90:1406-1415).  be: as in cast_f32."""
    import torch
    d, was_t = _dev(x)
    if be and not was_t:
        raise ValueError("big-endian results are for the device flow: pass a CUDA tensor")
    out = torch.empty((d.shape[0] - 1,) + tuple(d.shape[1:]), dtype=torch.float32, device=d.device)
    check(L.load().mpg_post_layer_mean_dev(C.c_void_p(d.data_ptr()), C.c_int(d.shape[0]), C.c_int64(d[0].numel()),
                                           C.c_void_p(out.data_ptr()), C.c_int(int(be)), _stream()))
    if be:
        out.mpg_be = True
    return out if was_t else out.cpu().numpy()


def p_top(p_hyd):
    """P_TOP (write_data.F90:1362-1371) as the float32 the file holds."""
    d, _ = _dev(p_hyd)
    v = C.c_double()
    check(L.load().mpg_post_ptop_dev(C.c_void_p(d.data_ptr()), C.c_int(d.shape[0]), C.c_int64(d[0].numel()), C.byref(v), _stream()))
    return np.float32(v.value)


def _is_cuda(x):
    return type(x).__module__.startswith("torch") and x.is_cuda


def _zeros_like_f32(x):
    import torch
    if type(x).__module__.startswith("torch"):
        return torch.zeros(x.shape, dtype=torch.float32, device=x.device)
    return np.zeros(x.shape, np.float32)


def output_fields(out, cfg, be=False):
    """interp_data's result (target name -> float64 array) -> ordered dict of what write_target_data puts in the file
    for those fields (float32).  Grid variables (XLAT, MAPFAC_*, ...) and time records are not part of the hot path.
    be: CUDA results big-endian (io_nc's device flow writes them to the file as they are)."""
    def cast_f32(x, scale=1.0, offset=0.0):
        return globals()["cast_f32"](x, scale, offset, be=be and _is_cuda(x))

    def layer_mean_f32(x):
        return globals()["layer_mean_f32"](x, be=be and _is_cuda(x))

    h = F.classify_hist(cfg.hist_2d, cfg.hist_3d, cfg.hist_soil, cfg.wrf_mod_vars)
    res = {}
    if "HGT" in out:
        res["HGT"] = cast_f32(out["HGT"])                                # :1150-1155
    if h.do_u_interp and "U" in out:
        res["U"] = cast_f32(out["U"])                                    # :1157-1172
    if h.do_v_interp and "V" in out:
        res["V"] = cast_f32(out["V"])                                    # :1175-1190
    two_d = [(n, t) for n, t in cfg.diag_list if cfg.interp_diag and out[t].ndim == 2] if cfg.diag_list else []
    for _, t in two_d + h.cons_2d + h.patch_2d + h.nstd_2d:              # field_write_2d (filled :584,643,678,711), :1245-1262
        if t in out:
            res[t] = cast_f32(out[t])
    for n, t in (cfg.diag_list if cfg.interp_diag else []):              # 3-D diag fields (REFL_10CM), :1266-1282
        if t in out and out[t].ndim == 3:
            res[t] = cast_f32(out[t])
    for _, t in h.soil:                                                  # :1285-1308
        res[t] = cast_f32(out[t])
    for _, t in h.nz_3d:                                                 # :1312-1382
        if cfg.wrf_mod_vars and t == "T":
            res[t] = cast_f32(out[t], offset=-300.0)
        else:
            res[t] = cast_f32(out[t])
        if cfg.wrf_mod_vars and t == "MUB":
            res["MU"] = _zeros_like_f32(out[t])
        if cfg.wrf_mod_vars and t == "P_HYD":
            res["P_TOP"] = p_top(out[t])
            res["PB"] = res[t]
    for _, t in h.nzp1_3d:                                               # :1386-1436
        if t == "PHB":
            res["Z_C"] = layer_mean_f32(out[t])
            res[t] = cast_f32(out[t], scale=9.81)
        else:
            res[t] = cast_f32(out[t])
        if cfg.wrf_mod_vars and t == "PHB":
            res["PH"] = _zeros_like_f32(out[t])
    for _, t in h.vert_3d:                                               # :1439-1463
        res[t] = cast_f32(out[t])
    if cfg.wrf_mod_vars and h.nz_3d:                                     # :1466-1475
        res["P"] = _zeros_like_f32(out[h.nz_3d[0][1]])
    return res
