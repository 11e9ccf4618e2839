"""Output epilogues (write_data.F90:1339-1475 as device kernels) against the oracle's numpy restatement: bit-exact,
every operation is a single correctly rounded float64 op followed by the float64 -> float32 conversion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cast_layer_mean_ptop_bit_exact(oracle, gpu_lib):
    import torch

    from mpassit_amd import post
    rng = np.random.default_rng(11)
    for shape in ((5, 37, 41), (3, 1, 7), (56, 64, 129)):
        x = rng.normal(300.0, 40.0, shape) * 10.0 ** rng.integers(-3, 4, shape)
        assert np.array_equal(post.cast_f32(x), oracle.post_cast(x))
        assert np.array_equal(post.cast_f32(x, offset=-300.0), oracle.post_cast(x, offset=-300.0))
        assert np.array_equal(post.cast_f32(x, scale=9.81), oracle.post_cast(x, scale=9.81))
        assert np.array_equal(post.layer_mean_f32(x), oracle.post_layer_mean(x))
        xt = torch.as_tensor(x, device="cuda")
        odd = xt.reshape(-1)[1:]                                      # 8-byte aligned only: scalar path
        assert np.array_equal(post.cast_f32(odd).cpu().numpy(), oracle.post_cast(x.reshape(-1)[1:]))
        assert post.cast_f32(xt).dtype == torch.float32 and post.cast_f32(xt).is_cuda
        # big-endian results (the bytes nf90_put_var stores in a classic file): the oracle's values, byte-reversed
        for got, want in ((post.cast_f32(xt, scale=9.81, offset=-300.0, be=True), oracle.post_cast(x, scale=9.81, offset=-300.0)),
                          (post.cast_f32(odd, be=True), oracle.post_cast(x.reshape(-1)[1:])),
                          (post.layer_mean_f32(xt, be=True), oracle.post_layer_mean(x))):
            assert got.mpg_be and got.cpu().numpy().tobytes() == want.astype(">f4").tobytes()
    p = np.abs(rng.normal(5.0e4, 2.0e4, (55, 30, 40)))
    p[-1] = rng.uniform(0.0, 6000.0, (30, 40))
    p[-1, :3] = 0.0                                                   # unmapped columns (regrid leaves 0.0) are skipped
    assert post.p_top(p) == oracle.post_ptop(p)
    p[-1] = 5.0                                                       # no column reaches 10: maxval of the whole field
    assert post.p_top(p) == oracle.post_ptop(p) == np.float32(p.max())
    neg = -np.abs(p)
    assert post.p_top(neg) == oracle.post_ptop(neg)                   # ordered-key reduction handles negative values


def test_output_fields_follow_the_writer(oracle, gpu_lib):
    """post.output_fields on a tiny wrf_mod_vars run: names, order and values of what write_target_data would write."""
    import torch

    from mpassit_amd import interp as I, post, regrid as R, workloads
    m, g, nz, _ = workloads.workload("tiny")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rng = np.random.default_rng(3)
    hist_2d = [("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW")]
    hist_3d = [("zgrid", "PHB"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"), ("pressure", "P_HYD"),
               ("rho", "MUB")]
    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=0, hgt=rng.uniform(0, 3000, m.nCells))
    inp.hist = {"xland": np.floor(rng.uniform(1, 3, m.nCells)), "skintemp": rng.uniform(250, 320, m.nCells),
                "snow": rng.uniform(0, 50, m.nCells), "zgrid": np.sort(rng.uniform(0, 2.0e4, (nz + 1, m.nCells)), axis=0),
                "theta": rng.uniform(280, 500, (nz, m.nCells)), "uReconstructZonal": rng.normal(0, 10, (nz, m.nCells)),
                "uReconstructMeridional": rng.normal(0, 10, (nz, m.nCells)),
                "pressure": -np.sort(-rng.uniform(2.0e3, 1.0e5, (nz, m.nCells)), axis=0), "rho": rng.uniform(0.1, 1.2, (nz, m.nCells))}
    cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_2d=hist_2d, hist_3d=hist_3d)
    out = I.interp_data(mesh, grid, g, inp, cfg)
    res = post.output_fields(out, cfg)
    assert list(res) == ["HGT", "U", "V", "SNOW", "TSK", "XLAND", "T", "P_HYD", "P_TOP", "PB", "MUB", "MU", "Z_C", "PHB", "PH", "P"]
    assert all(v.dtype == np.float32 for k, v in res.items() if k != "P_TOP")
    assert np.array_equal(res["T"], oracle.post_cast(out["T"], offset=-300.0))
    assert np.array_equal(res["PHB"], oracle.post_cast(out["PHB"], scale=9.81))
    assert np.array_equal(res["Z_C"], oracle.post_layer_mean(out["PHB"])) and res["Z_C"].shape == (nz, g.ny, g.nx)
    assert res["P_TOP"] == oracle.post_ptop(out["P_HYD"]) and np.array_equal(res["PB"], res["P_HYD"])
    assert res["U"].shape == (nz, g.ny, g.nx + 1) and res["V"].shape == (nz, g.ny + 1, g.nx)
    for z in ("MU", "PH", "P"):
        assert not res[z].any()
    assert res["PH"].shape == (nz + 1, g.ny, g.nx) and res["P"].shape == (nz, g.ny, g.nx)
    # the device-resident route returns CUDA float32 tensors with the same bits
    out_t = {k: torch.as_tensor(v, device="cuda") for k, v in out.items()}
    res_t = post.output_fields(out_t, cfg)
    for k in res:
        if k != "P_TOP":
            assert res_t[k].is_cuda and np.array_equal(res_t[k].cpu().numpy(), res[k])
    mesh.destroy()
    grid.destroy()
