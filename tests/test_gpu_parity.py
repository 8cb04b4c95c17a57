"""GPU parity tests: the HIP render path (through the C ABI) against the golden vectors made from the reference and
against the CPU oracle.  Run on an MI355X with  pytest -m gpu.

Tolerances (fp32): `scaled_err` = max |a-b| / max(|b|, rms(b)) must stay below 1e-4 for rgb/sigma/pixels/depth
(north-star gate); geometry (sample positions, jittered depths, looked-up features) is bit-exact; integer outputs
(inds, sort_idx) are bit-exact except inside a guard band around a cdf entry / between near-equal depths, where a
1-ulp difference upstream legitimately flips the decision -- the excluded fraction is asserted tiny.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_NAMES as ALL_GOLDEN, SMALL_GOLDEN as ALL_SMALL, DROP_GOLDEN, NOT_ON_GPU_YET, NO_BACKWARD_YET, scaled_err

GOLDEN_NAMES = [n for n in ALL_GOLDEN if n not in NOT_ON_GPU_YET]
SMALL_GOLDEN = [n for n in ALL_SMALL if n not in NOT_ON_GPU_YET]

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch.device("cuda:0")


def G(x, dev):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def make_generator(g, dev):
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    m = g.meta
    dp = m.get("drop_out", 0)
    if m["variant"] == "TALLSIREN":
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["C"], input_dim=3, output_dim=4, hidden_dim=m["H"], drop_out=dp)
    elif m["has_global"]:
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["Z"], input_dim=m.get("input_dim", m["C"]), output_dim=4, hidden_dim=m["H"], drop_out=dp)
    else:
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["C"], input_dim=m["C"], output_dim=4, hidden_dim=m["H"], drop_out=dp)
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.d.files if k.startswith("param/")}
    gen.load_state_dict(sd, strict=True)
    gen.to(dev)
    gen.set_device(dev)
    gen.eval()
    return gen


SPLIT_FIXTURES = [n for n in GOLDEN_NAMES if n.startswith(("short_fg", "tall_fg", "double_fg", "single_dg", "short_f_", "tall_dgx", "short_pyrmd", "short_fres", "tall_dres", "tallsiren"))]


@pytest.mark.parametrize("name", SPLIT_FIXTURES)
def test_split_precision(golden, dev, name):
    """precision = "fp16x3": every fp32 product evaluated as three fp16 MFMAs (two-way split of both operands).  Same gates
    as the fp32 path: geometry bit-exact, rgb / sigma / image within 1e-4 (scaled), merge order bit-exact."""
    g = golden(name)
    m = g.meta
    gen = make_generator(g, dev)
    gen.siren.precision = "fp16x3"
    z, _, _ = make_z(g, dev)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]:
        rng["fine_z"] = G(g["fine_z"], dev)
    aux = {}
    with torch.no_grad():
        pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                            clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                            _rng=rng, _aux=aux)
    aux = {k: v.cpu().numpy() for k, v in aux.items()}
    assert np.array_equal(aux["coarse_z"], g["coarse_z"])
    assert scaled_err(aux["coarse_rgb_sigma"][..., :3], g["coarse_rgb_sigma"][..., :3]) < TOL
    assert scaled_err(aux["coarse_rgb_sigma"][..., 3], g["coarse_rgb_sigma"][..., 3]) < TOL
    if m["hierarchical"]:
        assert scaled_err(aux["fine_rgb_sigma"][..., :3], g["fine_rgb_sigma"][..., :3]) < TOL
        assert scaled_err(aux["fine_rgb_sigma"][..., 3], g["fine_rgb_sigma"][..., 3]) < TOL
        assert np.array_equal(aux["sort_idx"], g["sort_idx"].astype(np.int32))
    # the image integrates 2S samples through exp(-delta * sigma) with the fixtures' 40x density head: allow 2e-4 there
    assert scaled_err(pixels.cpu().numpy(), g["pixels"]) < 2 * TOL
    assert scaled_err(depth.cpu().numpy(), g["depth"]) < 2 * TOL


@pytest.mark.parametrize("name", SPLIT_FIXTURES)
def test_single_pass_fp16(golden, dev, name):
    """precision = "fp16": plain fp16 products with fp32 accumulation (one MFMA per 16 k-values, operands rounded to nearest,
    weights pre-scaled by a power of two per matrix) -- the arithmetic BASELINE config 5 names ("bf16 SIREN on MFMA"; fp16 has
    the same MFMA rate and 8x less operand rounding) and the class of the reference's own GPU path (fp16 autocast).  It is NOT
    a parity path and does not claim the 1e-4 gate: a w0 = 30 SIREN amplifies the 5e-4 operand rounding ~5x per layer, and the
    fixtures' x40 density head on top.  Stated tolerance: rgb / sigma of both passes within 1e-1 (scaled; measured 2e-4 on the
    residual families .. 5e-2 on the 4- and 8-layer FiLM ones -- the reference under its own fp16 autocast: 1.8e-1,
    test_single_pass_fp16_vs_reference_autocast), mean |pixel error| below 3e-2 with the reference's fine depths forced.
    Geometry and the merge order stay bit-exact (same position / lookup code)."""
    g = golden(name)
    m = g.meta
    gen = make_generator(g, dev)
    gen.siren.precision = "fp16"
    z, _, _ = make_z(g, dev)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]:
        rng["fine_z"] = G(g["fine_z"], dev)
    aux = {}
    with torch.no_grad():
        pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                            clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                            _rng=rng, _aux=aux)
    aux = {k: v.cpu().numpy() for k, v in aux.items()}
    assert np.array_equal(aux["coarse_z"], g["coarse_z"])
    e = [rgb_sigma_err(aux["coarse_rgb_sigma"], g["coarse_rgb_sigma"])]
    if m["hierarchical"]:
        e.append(rgb_sigma_err(aux["fine_rgb_sigma"], g["fine_rgb_sigma"]))
        assert np.array_equal(aux["sort_idx"], g["sort_idx"].astype(np.int32))
    mp = float(np.abs(pixels.cpu().numpy() - g["pixels"]).mean())
    md = float(np.abs(depth.cpu().numpy() - g["depth"]).mean())
    print(f"{name} [fp16]: rgb/sigma scaled_err (coarse, fine) = {['%.1e' % v for v in e]}, mean |pixel err| {mp:.1e}, mean |depth err| {md:.1e}")
    assert max(e) < 1e-1, e
    assert mp < 3e-2 and md < 3e-2, (mp, md)


def test_single_pass_fp16_vs_reference_autocast(golden, dev):
    """The single-pass fp16 forward against what the reference's own GPU numerics do to the same field: on `short_fg_small` the
    reference's coarse rgb / sigma under fp16 autocast (emulated on the CPU: nn.Linear in fp16) differ from its fp32 values by
    more than the HIP fp16 kernel does (which keeps lookup, FiLM and sine in fp32 and rounds only the MFMA operands)."""
    from oracle import render_oracle as O
    g = golden("short_fg_small")
    m = g.meta
    T = lambda x: None if x is None else torch.from_numpy(np.asarray(x))
    pts = T(g["coarse_points"]).reshape(m["B"], -1, 3)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        amp, _ = O.field_eval(O.FIELD_SPECS[m["variant"]], {k: T(v) for k, v in g.params().items()}, T(g["feature_volume"]), T(g["global_feature"]), pts)
    amp = amp.float().numpy().reshape(g["coarse_rgb_sigma"].shape)
    gen = make_generator(g, dev)
    gen.siren.precision = "fp16"
    z, _, _ = make_z(g, dev)
    with torch.no_grad():
        out = gen.siren(G(g["coarse_points"], dev).reshape(m["B"], -1, 3), z, m["R"], m["S"]).cpu().numpy().reshape(g["coarse_rgb_sigma"].shape)
    e_amp, e_hip = rgb_sigma_err(amp, g["coarse_rgb_sigma"]), rgb_sigma_err(out, g["coarse_rgb_sigma"])
    print(f"reference under fp16 autocast vs its fp32: {e_amp:.2e} | HIP single-pass fp16 vs reference fp32: {e_hip:.2e}")
    assert e_hip < e_amp


def make_z(g, dev, requires_grad=False):
    """(z as the generator takes it, list of volume leaves, global feature leaf or None)."""
    vols = g.volumes()
    leaves = [G(v, dev).requires_grad_(requires_grad) for v in (vols if isinstance(vols, list) else [vols])]
    fv = leaves if isinstance(vols, list) else leaves[0]
    if g.meta["has_global"]:
        glob = G(g["global_feature"], dev).requires_grad_(requires_grad)
        return (fv, glob), leaves, glob
    return fv, leaves, None


def test_library_loaded_and_no_fallback(dev):
    import cnerf_amd
    assert cnerf_amd._lib.lib().cnerf_abi_version() == cnerf_amd._lib.ABI_VERSION
    from cnerf_amd.generators import ImplicitGenerator3d
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 16, 32, 4, 64)
    with pytest.raises(cnerf_amd._lib.CnerfError):   # CPU tensors are refused, never computed on the host
        gen.siren(torch.zeros(1, 32, 3), (torch.zeros(1, 32, 4, 4, 4), torch.zeros(1, 16)))


def test_channel_last_round_trip(dev):
    import cnerf_amd
    x = torch.randn(2, 32, 9, 9, 9, device=dev)
    cl = cnerf_amd.ops.channel_last(x)
    assert torch.equal(cl, x.permute(0, 2, 3, 4, 1).contiguous())
    assert torch.equal(cnerf_amd.ops.channel_first(cl), x)


@pytest.mark.parametrize("name", SMALL_GOLDEN)
def test_trilinear_lookup_bit_exact(golden, dev, name):
    """Unfused gather kernel == F.grid_sample of the reference, bit for bit (same corner order, no fma)."""
    import cnerf_amd
    g = golden(name)
    gen = make_generator(g, dev)
    B = g.meta["B"]
    pts = G(g["coarse_points"], dev).reshape(B, -1, 3)
    cl = cnerf_amd.ops.channel_last(G(g["feature_volume"], dev))
    feat = cnerf_amd.ops.gather_features(gen.siren, cl, pts).cpu().numpy()
    assert np.array_equal(feat, g["coarse_feat"][..., :32])      # level 0 of a pyramid / the single volume


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_field_network(golden, dev, name):
    """siren sub-API at the reference's own sample points: rgb and sigma within 1e-4 (scaled)."""
    g = golden(name)
    if "coarse_points" not in g:
        pytest.skip("fixture stores no sample points")
    m = g.meta
    gen = make_generator(g, dev)
    z, _, _ = make_z(g, dev)
    pts = G(g["coarse_points"], dev).reshape(m["B"], -1, 3)
    with torch.no_grad():
        out = gen.siren(pts, z, m["R"], m["S"]).cpu().numpy().reshape(g["coarse_rgb_sigma"].shape)
    ref = g["coarse_rgb_sigma"]
    assert scaled_err(out[..., :3], ref[..., :3]) < TOL
    assert scaled_err(out[..., 3], ref[..., 3]) < TOL


@pytest.mark.parametrize("name", SMALL_GOLDEN)
def test_composite_stage(golden, dev, name):
    """fancy_integration stage fed with the reference's own rgb_sigma: weights within 1e-5."""
    import cnerf_amd
    g = golden(name)
    m = g.meta
    if not m["hierarchical"]:
        pytest.skip("no coarse weights stored")
    B, P, S = g["coarse_z"].shape
    eps = g.get("eps_coarse")
    rgb, dist, w = cnerf_amd.ops.composite(G(g["coarse_rgb_sigma"], dev).reshape(B * P, S, 4), G(g["coarse_z"], dev).reshape(B * P, S),
                                           G(eps, dev).reshape(B * P, S) if eps is not None else None, m["noise"], m["clamp"])
    assert scaled_err(w.cpu().numpy().reshape(B, P, S), g["coarse_weights"]) < 1e-5
    assert torch.all(w.sum(-1) <= 1 + 1e-5)


def flips_outside_band(cdf, u, mine, ref, band):
    bad = mine != ref
    robust = (np.abs(u[..., :, None] - cdf[..., None, :]) > band).all(-1)
    return int((bad & robust).sum()), float(bad.mean())


def bin_mass(cdf, inds):
    """cdf[above] - cdf[below] of each draw: the denominator of the inverse-CDF interpolation.  Where it is tiny the
    interpolated depth is ill-conditioned: a 1-ulp change of a cdf entry (the reference's own fp32 `sum` changes by
    that much between AVX2 and AVX-512 hosts) moves the depth by ulp/den of a bin width."""
    S = cdf.shape[-1] + 1
    below = np.clip(inds.astype(np.int64) - 1, 0, None)
    above = np.clip(inds.astype(np.int64), None, S - 2)
    return np.take_along_axis(cdf, above, -1) - np.take_along_axis(cdf, below, -1)


@pytest.mark.parametrize("name", SMALL_GOLDEN)
def test_resample_stage(golden, dev, name):
    """sample_pdf stage fed with the reference's own weights: bin indices bit-exact (outside a 2e-6 guard band
    around cdf entries), fine depths within 1e-5."""
    import cnerf_amd
    g = golden(name)
    if not g.meta["hierarchical"]:
        pytest.skip("not hierarchical")
    B, P, S = g["coarse_z"].shape
    fine, inds, cdf = cnerf_amd.ops.resample(G(g["coarse_z"], dev).reshape(B * P, S), G(g["coarse_weights"], dev).reshape(B * P, S),
                                             G(g["u_fine"], dev).reshape(B * P, S))
    cdf = cdf.cpu().numpy().reshape(B, P, S - 1)
    assert scaled_err(cdf, g["cdf"]) < 1e-6
    hard, frac = flips_outside_band(g["cdf"], g["u_fine"], inds.cpu().numpy().reshape(B, P, S), g["inds"].astype(np.int32), 2e-6)
    assert hard == 0 and frac < 1e-3
    same = inds.cpu().numpy().reshape(B, P, S) == g["inds"]
    fz = fine.cpu().numpy().reshape(B, P, S)
    well = same & (bin_mass(g["cdf"], g["inds"]) > 1e-2)
    assert well.mean() > 0.9
    assert scaled_err(fz[well], g["fine_z"][well]) < 1e-5
    # everywhere else the draw still lands inside the same bin
    assert np.abs(fz[same] - g["fine_z"][same]).max() < 2 * (g.meta["ray_end"] - g.meta["ray_start"]) / (S - 1)


def _render_with(g, dev, forced, precision="fp32"):
    m = g.meta
    gen = make_generator(g, dev)
    gen.siren.precision = precision
    z, _, _ = make_z(g, dev)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if forced:
        rng["fine_z"] = G(g["fine_z"], dev)
    aux = {}
    with torch.no_grad():
        pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"],
                            m["hierarchical"], clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"],
                            last_back=m["last_back"], _rng=rng, _aux=aux, batch_size=7, generator={"x": 1})
    torch.cuda.synchronize()
    return pixels.cpu().numpy(), depth.cpu().numpy(), {k: v.cpu().numpy() for k, v in aux.items()}


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_render_teacher_forced(golden, dev, name):
    """ImplicitGenerator3d.forward with the reference's random draws AND its resampled depths injected: every stage
    downstream of the resampling sees the reference's own sample positions, so the whole image must agree to 1e-4
    and the merge permutation bit for bit.  (Without forcing, the 1e-5-level rounding of the coarse densities moves
    the fine depths by ~1e-6, and the random-init SIREN on a random volume turns that into O(1e-2) colour changes:
    the field is chaotic in position, see test_render_free_running.)"""
    g = golden(name)
    m = g.meta
    pixels, depth, aux = _render_with(g, dev, forced=m["hierarchical"])
    if "coarse_points" in g:
        assert np.array_equal(aux["coarse_points"], g["coarse_points"])        # geometry: bit-exact
    assert np.array_equal(aux["coarse_z"], g["coarse_z"])
    assert scaled_err(aux["coarse_rgb_sigma"][..., :3], g["coarse_rgb_sigma"][..., :3]) < TOL
    assert scaled_err(aux["coarse_rgb_sigma"][..., 3], g["coarse_rgb_sigma"][..., 3]) < TOL
    if m["hierarchical"]:
        if "fine_points" in g:
            assert np.array_equal(aux["fine_points"], g["fine_points"])
        assert scaled_err(aux["fine_rgb_sigma"][..., :3], g["fine_rgb_sigma"][..., :3]) < TOL
        assert scaled_err(aux["fine_rgb_sigma"][..., 3], g["fine_rgb_sigma"][..., 3]) < TOL
        assert np.array_equal(aux["sort_idx"], g["sort_idx"].astype(np.int32))  # merge order: bit-exact
    if "final_weights" in g:
        # weights multiply up to 2S factors exp(-delta*sigma): the 1e-5-level density rounding accumulates
        assert scaled_err(aux["final_weights"], g["final_weights"]) < 5e-4
    assert scaled_err(pixels, g["pixels"]) < TOL
    assert scaled_err(depth, g["depth"]) < TOL


@pytest.mark.parametrize("name", [n for n in GOLDEN_NAMES if n != "short_fg_nohier"])
def test_render_free_running(golden, dev, name):
    """The same call without forcing: coarse pass as above; resampling decisions bit-exact outside a guard band of
    1e-4 around the cdf entries (the coarse weights carry the 1e-5-level rounding of the densities); depths of draws
    that fall into bins of non-negligible mass (> 1e-2) within 3e-3 of a bin width; images close on average."""
    g = golden(name)
    m = g.meta
    pixels, depth, aux = _render_with(g, dev, forced=False)
    assert scaled_err(aux["coarse_weights"], g["coarse_weights"]) < 5e-4
    hard, frac = flips_outside_band(aux["cdf"], g["u_fine"], aux["inds"], g["inds"].astype(np.int32), 1e-4)
    assert hard == 0, "bin index differs outside the guard band"
    assert frac < 5e-3
    well = (aux["inds"] == g["inds"]) & (bin_mass(aux["cdf"], g["inds"]) > 1e-2)
    assert well.mean() > 0.8
    binw = (m["ray_end"] - m["ray_start"]) / (m["S"] - 1)
    assert np.abs(aux["fine_z"] - g["fine_z"])[well].max() < 3e-3 * binw
    assert np.abs(pixels - g["pixels"]).mean() < 2e-3
    assert np.abs(depth - g["depth"]).mean() < 2e-3


def err_stats(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel()
    return {"mean": float(d.mean()), "p99": float(np.quantile(d, 0.99)), "p999": float(np.quantile(d, 0.999)), "max": float(d.max())}


def oracle_free_running(g, dtype):
    """The oracle on a fixture's inputs and draws WITHOUT forcing anything, in float32 or float64."""
    from oracle import render_oracle as O
    m = g.meta
    T = lambda x: None if x is None else torch.from_numpy(np.asarray(x)).to(dtype)
    vols = g.volumes()
    fvol = [T(v) for v in vols] if isinstance(vols, list) else T(vols)
    torch.set_default_dtype(dtype)
    try:
        with torch.no_grad():
            return O.render(m["variant"], {k: T(v) for k, v in g.params().items()}, fvol, T(g.get("global_feature")), T(g["cam2worlds"]),
                            m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True, m["clamp"], m["noise"], m["white_back"],
                            m["last_back"], T(g["u_strat"]), T(g.get("eps_coarse")), T(g.get("u_fine")), T(g.get("eps_final")))
    finally:
        torch.set_default_dtype(torch.float32)


def free_running_floor(g):
    """How far the reference's OWN arithmetic is from exact on a free-running render: the fp32 oracle (bit-identical to the
    reference on the fixtures, tests/test_oracle_golden.py) against the same oracle in float64 -- identical inputs, draws
    and algorithm, only the rounding differs.  Error statistics of pixels and depth."""
    a, b = oracle_free_running(g, torch.float32), oracle_free_running(g, torch.float64)
    return err_stats(a.pixels, b.pixels), err_stats(a.depth, b.depth)


HIER_FIXTURES = [n for n in GOLDEN_NAMES if n not in ("short_fg_nohier", "short_fg_smooth")]


@pytest.mark.parametrize("name", HIER_FIXTURES)
def test_free_running_noise_floor(golden, dev, name):
    """Evidence behind the teacher forcing of the image-level tests.  A free-running render (nothing forced) of the HIP path
    differs from the reference's by MORE than 1e-4 on these fixtures -- and so does the reference from itself once only the
    rounding changes: its fp32 result against the same algorithm in float64 (`free_running_floor`) is off by 1e-4 .. 1e-1
    on single pixels, because the 1e-7-level rounding of the coarse densities moves resampled depths that fall into
    near-empty importance bins, and the random-init SIREN turns the shift into a colour change.  The HIP path must sit
    inside that floor: mean pixel / depth error no more than 3x, 99th percentile no more than 4x the reference's own
    fp32-vs-fp64 figures (+2e-6 absolute: where the floor is 1e-7 the two fp32 implementations differ by their summation
    order, not by conditioning).  Worst pixels are single rays on either side (printed, not asserted; measured table:
    profiles/r02_parity_report.md)."""
    g = golden(name)
    f_px, f_dp = free_running_floor(g)
    for prec in (("fp32", "fp16x3") if name in SPLIT_FIXTURES else ("fp32",)):
        pixels, depth, aux = _render_with(g, dev, forced=False, precision=prec)
        h_px, h_dp = err_stats(pixels, g["pixels"]), err_stats(depth, g["depth"])
        print(f"{name} [{prec}]: pixels HIP-vs-reference {h_px} | reference fp32-vs-fp64 {f_px}")
        print(f"{name} [{prec}]: depth  HIP-vs-reference {h_dp} | reference fp32-vs-fp64 {f_dp}")
        for h, f in ((h_px, f_px), (h_dp, f_dp)):
            assert h["mean"] <= 3 * f["mean"] + 2e-6, (prec, h, f)
            assert h["p99"] <= 4 * f["p99"] + 2e-6, (prec, h, f)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_free_running_well_conditioned(golden, dev, precision):
    """`short_fg_smooth`: a fixture of the reference where free-running parity is a fair question -- smooth feature volume,
    mild head, softplus density, so every importance bin carries mass and the reference's own fp32-vs-fp64 distance is
    ~1e-6.  NOTHING is forced here: the whole hierarchical render (coarse pass, resampling, fine pass, merge, composite)
    runs on its own and the image, the depth map, the resampled depths and both rgb/sigma tensors agree with the reference
    to the north-star tolerance; bins and merge order bit for bit."""
    g = golden("short_fg_smooth")
    m = g.meta
    f_px, f_dp = free_running_floor(g)
    assert f_px["max"] < 2e-5 and f_dp["max"] < 2e-5          # the scene is well conditioned for the reference itself
    gen = make_generator(g, dev)
    gen.siren.precision = precision
    z, _, _ = make_z(g, dev)
    rng = {k: G(g[k], dev) for k in ("u_strat", "u_fine")}
    aux = {}
    with torch.no_grad():
        pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True,
                            clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                            _rng=rng, _aux=aux)
    aux = {k: v.cpu().numpy() for k, v in aux.items()}
    assert np.array_equal(aux["coarse_points"], g["coarse_points"])
    assert np.array_equal(aux["inds"], g["inds"].astype(np.int32))
    assert np.array_equal(aux["sort_idx"], g["sort_idx"].astype(np.int32))
    assert scaled_err(aux["fine_z"], g["fine_z"]) < 3e-5
    assert rgb_sigma_err(aux["coarse_rgb_sigma"], g["coarse_rgb_sigma"]) < TOL
    # the fine pass evaluates the field at the run's OWN resampled depths, which differ from the reference's by ~1e-5 (above):
    # point-wise its rgb / sigma carry that displacement times the field's slope along the ray (measured 1e-4 .. 3e-4)
    assert rgb_sigma_err(aux["fine_rgb_sigma"], g["fine_rgb_sigma"]) < 1e-3
    e_p, e_d = scaled_err(pixels.cpu().numpy(), g["pixels"]), scaled_err(depth.cpu().numpy(), g["depth"])
    print(f"short_fg_smooth free-running [{precision}]: pixels {e_p:.2e} depth {e_d:.2e}; survey-metric pass "
          f"{survey_metric_pass(pixels.cpu().numpy(), g['pixels']):.4f}")
    assert e_p < TOL and e_d < TOL
    assert np.abs(pixels.cpu().numpy() - g["pixels"]).max() < 1e-4      # absolute, too: |pixels| <= 0.2 here


GRAD_FIXTURES = [n for n in GOLDEN_NAMES if n.endswith("_small") or n in ("short_fg_nohier", "short_fg_s40")]
RES_FIXTURES = set(NO_BACKWARD_YET)      # (residual-block networks have a backward too)


def reference_grad_noise_floor(g):
    """scaled_err between the reference's fp32 gradients (fixture) and the same gradients evaluated in float64 by the
    oracle at IDENTICAL sample positions (fine depths forced): how much of a gradient is fp32 rounding noise.  The
    4-layer FiLM networks amplify forward rounding by ~1e2..1e3 per gradient entry; 2-layer ones by ~10."""
    from oracle import render_oracle as O
    m = g.meta
    T = lambda x: None if x is None else torch.from_numpy(np.asarray(x)).double()
    params = {k: T(v).requires_grad_(True) for k, v in g.params().items()}
    vols = g.volumes()
    vleaves = [T(v).requires_grad_(True) for v in (vols if isinstance(vols, list) else [vols])]
    fvol = vleaves if isinstance(vols, list) else vleaves[0]
    glob = T(g.get("global_feature"))
    if glob is not None:
        glob.requires_grad_(True)
    torch.set_default_dtype(torch.float64)
    try:
        out = O.render(m["variant"], params, fvol, glob, T(g["cam2worlds"]), m["R"], m["fov"], m["ray_start"], m["ray_end"],
                       m["S"], m["hierarchical"], m["clamp"], m["noise"], m["white_back"], m["last_back"], T(g["u_strat"]),
                       T(g.get("eps_coarse")), T(g.get("u_fine")), T(g.get("eps_final")),
                       forced_fine_z=T(g.get("fine_z")) if m["hierarchical"] else None, **g.oracle_dropout(T))
    finally:
        torch.set_default_dtype(torch.float32)
    loss = out.pixels.square().mean() + out.depth.mean()
    leaves = vleaves + ([glob] if glob is not None else []) + list(params.values())
    grads = torch.autograd.grad(loss, leaves)
    floor = {}
    for li in range(len(vleaves)):
        key = "grad_feature_volume" + (f"_l{li}" if li else "")
        floor["feature_volume" + (f"_l{li}" if li else "")] = scaled_err(g[key], grads[li].numpy())
    i = len(vleaves)
    if glob is not None:
        floor["global_feature"] = scaled_err(g["grad_global_feature"], grads[i].numpy())
        i += 1
    for (k, _), gg in zip(params.items(), grads[i:]):
        floor["siren." + k] = scaled_err(g["grad/siren." + k], gg.numpy())
    return floor


@pytest.mark.parametrize("name", [n for n in GRAD_FIXTURES if n in SPLIT_FIXTURES])
def test_backward_split_precision(golden, dev, name):
    """The same gradient gate with precision = "fp16x3": forward and the activation-storing re-run of the backward in the
    split-precision kernel (hardware sine / cosine), gradient chain and weight-gradient reduction in fp32."""
    test_backward_teacher_forced(golden, dev, name, precision="fp16x3")


HALF_BACKWARD_FIXTURES = [n for n in GRAD_FIXTURES if n.startswith(("short_fg", "tall_fg", "double_fg", "single_dg", "short_f_", "tall_dgx", "short_pyrmd",
                                                                     "short_fres", "tall_dres", "tallsiren"))]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def reference_autocast_grads(g):
    """The gradients the reference's OWN training numerics give on a fixture: its GPU trainer wraps the whole step in
    torch.cuda.amp.autocast (utils.py:643-711: every nn.Linear of the field network in fp16, forward and backward).  Emulated
    on the CPU by running the oracle (same ATen op sequence) under torch.autocast("cpu", float16), fine depths forced."""
    from oracle import render_oracle as O
    m = g.meta
    T = lambda x: None if x is None else torch.from_numpy(np.asarray(x))
    params = {k: T(v).clone().requires_grad_(True) for k, v in g.params().items()}
    fv, gl = T(g["feature_volume"]).clone().requires_grad_(True), T(g["global_feature"]).clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.float16):
        out = O.render(m["variant"], params, fv, gl, T(g["cam2worlds"]), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True, m["clamp"],
                       m["noise"], m["white_back"], m["last_back"], T(g["u_strat"]), T(g.get("eps_coarse")), T(g.get("u_fine")), T(g.get("eps_final")),
                       forced_fine_z=T(g["fine_z"]))
        loss = out.pixels.float().square().mean() + out.depth.float().mean()
    grads = torch.autograd.grad(loss, [fv, gl] + list(params.values()))
    names = ["feature_volume", "global_feature"] + ["siren." + k for k in params]
    return {k: v.float().numpy() for k, v in zip(names, grads)}


@pytest.mark.parametrize("name", HALF_BACKWARD_FIXTURES)
def test_backward_half_precision(golden, dev, name):
    """backward_precision = "fp16": gradient chain and weight-gradient reductions on the fp16 MFMA -- fp16 operands scaled by
    powers of two, fp32 sums; the FORWARD stays fp32-accurate (fp16x3) and is re-run to keep its activations as fp16 tile blocks.
    Against the reference's fp32 CPU autograd every gradient tensor is within 2e-3 in relative L2 norm (measured 1e-4 .. 9e-4,
    profiles/r02_grad_report.md) and within max(5e-2, 2.5 x the reference's fp32-vs-fp64 distance) in the max-norm metric of the
    fp32 test (measured <= 2.5e-2: entries far below a tensor's rms carry the operand rounding, 2^-11, of the terms that
    cancelled in them).  For scale: the reference's own GPU training numerics -- fp16 autocast of forward AND backward -- sit
    20 - 30 % (relative L2) from its fp32 gradients on these fixtures (test_half_precision_backward_vs_reference_autocast)."""
    g = golden(name)
    got, ref = _hip_gradients(g, dev, precision="fp16x3", backward_precision="fp16")
    floor = reference_grad_noise_floor(g)
    for k in ref:
        assert rel_l2(got[k], ref[k]) < 2e-3, (k, rel_l2(got[k], ref[k]))
        assert scaled_err(got[k], ref[k]) < max(5e-2, 2.5 * floor[k]), (k, scaled_err(got[k], ref[k]))


def test_half_precision_backward_vs_reference_autocast(golden, dev):
    """How the half-precision backward compares with what the reference itself trains with: on `short_fg_small`, the reference
    under fp16 autocast (its GPU numerics, emulated on the CPU) is > 5 % away from its fp32 gradients on every tensor of the
    field network, the HIP fp16 backward at least 50 x closer."""
    g = golden("short_fg_small")
    got, ref = _hip_gradients(g, dev, precision="fp16x3", backward_precision="fp16")
    amp = reference_autocast_grads(g)
    for k in ref:
        if k in amp and "final_layer.bias" not in k:
            a, h = rel_l2(amp[k], ref[k]), rel_l2(got[k], ref[k])
            print(f"{k:40s} reference autocast vs fp32 {a:.2e} | HIP fp16 backward vs fp32 {h:.2e}")
            assert a > 5e-2 and h < a / 50, (k, a, h)


def _hip_gradients(g, dev, precision, backward_precision):
    """(HIP gradients, reference gradients) of the fixture's loss, keyed like reference_grad_noise_floor's result."""
    m = g.meta
    gen = make_generator(g, dev)
    gen.siren.precision = precision
    gen.siren.backward_precision = backward_precision
    gen.train()
    z, vleaves, glob = make_z(g, dev, requires_grad=True)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]:
        rng["fine_z"] = G(g["fine_z"], dev)
    pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                        clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng)
    (pixels.square().mean() + depth.mean()).backward()
    got, ref = {}, {}
    for li, leaf in enumerate(vleaves):
        sfx = f"_l{li}" if li else ""
        got["feature_volume" + sfx], ref["feature_volume" + sfx] = leaf.grad.cpu().numpy(), g["grad_feature_volume" + sfx]
    if glob is not None:
        got["global_feature"], ref["global_feature"] = glob.grad.cpu().numpy(), g["grad_global_feature"]
    for k, p in gen.named_parameters():
        got[k], ref[k] = p.grad.cpu().numpy(), g["grad/" + k]
    return got, ref


@pytest.mark.parametrize("name", [n for n in GRAD_FIXTURES if n not in RES_FIXTURES])
def test_backward_teacher_forced(golden, dev, name, precision="fp32", backward_precision="fp32"):
    """Gradients of  pixels.square().mean() + depth.mean()  w.r.t. every field parameter, the mapping network, the feature
    volume and the global feature, against the reference's autograd (stored in the fixture), with the reference's fine
    depths forced (see test_render_teacher_forced).
    Tolerance per tensor: max(2e-3, 2.5 x the reference's own fp32-vs-fp64 discrepancy on that tensor) in the metric
    max|a-b| / max(|b|, rms(b)) -- i.e. the HIP gradient must be as close to the reference as the reference is to exact
    arithmetic (measured: HIP error <= the reference's own noise on every tensor)."""
    g = golden(name)
    m = g.meta
    gen = make_generator(g, dev)
    gen.siren.precision = precision
    gen.siren.backward_precision = backward_precision
    gen.train()
    z, vleaves, glob = make_z(g, dev, requires_grad=True)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]:
        rng["fine_z"] = G(g["fine_z"], dev)
    pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                        clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                        _rng=rng)
    loss = pixels.square().mean() + depth.mean()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    floor = reference_grad_noise_floor(g)
    tol = lambda k: max(2e-3, 2.5 * floor[k])
    for li, leaf in enumerate(vleaves):
        sfx = f"_l{li}" if li else ""
        assert scaled_err(leaf.grad.cpu().numpy(), g["grad_feature_volume" + sfx]) < tol("feature_volume" + sfx), sfx
    if glob is not None:
        assert scaled_err(glob.grad.cpu().numpy(), g["grad_global_feature"]) < tol("global_feature")
    ref = {k[len("grad/"):]: g[k] for k in g.d.files if k.startswith("grad/")}
    for k, p in gen.named_parameters():
        assert p.grad is not None, k
        assert scaled_err(p.grad.cpu().numpy(), ref[k]) < tol(k), k


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_render_matches_oracle_random_inputs(dev, precision):
    """Fresh seeded inputs (not a stored fixture): HIP path vs the CPU oracle on identical rays and draws."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    from oracle import render_oracle as O
    torch.manual_seed(123)
    np.random.seed(123)
    B, R, S, V, H, Z = 2, 24, 24, 20, 128, 64
    gen = ImplicitGenerator3d("SHORTSIREN_FG", Z, 32, 4, H)
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 30
    fvol, glob = torch.randn(B, 32, V, V, V) * 0.5, torch.randn(B, Z)
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y")
    rng = {"u_strat": torch.rand(B, R * R, S), "eps_coarse": torch.randn(B, R * R, S), "u_fine": torch.rand(B, R * R, S),
           "eps_final": torch.randn(B, R * R, 2 * S)}
    params = {k: v.detach() for k, v in gen.siren.state_dict().items()}
    ref = O.render("SHORTSIREN_FG", params, fvol, glob, cam, R, 49.13, 0.25, 1.95, S, True, "softplus", 0.3, True, False,
                   rng["u_strat"], rng["eps_coarse"], rng["u_fine"], rng["eps_final"])
    gen.to(dev)
    gen.set_device(dev)
    gen.siren.precision = precision
    aux = {}
    with torch.no_grad():
        pixels, depth = gen((fvol.to(dev), glob.to(dev)), cam.to(dev), R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus",
                            nerf_noise=0.3, white_back=True, _rng={k: v.to(dev) for k, v in rng.items()}, _aux=aux)
    print(precision, "coarse rgb_sigma scaled_err", scaled_err(aux["coarse_rgb_sigma"].cpu().numpy(), ref.aux["coarse_rgb_sigma"].numpy()))
    assert torch.equal(aux["coarse_points"].cpu(), ref.aux["coarse_points"])
    assert scaled_err(aux["coarse_rgb_sigma"].cpu().numpy(), ref.aux["coarse_rgb_sigma"].numpy()) < TOL
    assert (aux["inds"].cpu() == ref.aux["inds"]).float().mean() > 0.995
    # second call with the oracle's fine depths forced: the whole image agrees
    rng_f = {k: v.to(dev) for k, v in rng.items()}
    rng_f["fine_z"] = ref.aux["fine_z"].to(dev)
    with torch.no_grad():
        pixels, depth = gen((fvol.to(dev), glob.to(dev)), cam.to(dev), R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus",
                            nerf_noise=0.3, white_back=True, _rng=rng_f, _aux=aux)
    assert torch.equal(aux["fine_points"].cpu(), ref.aux["fine_points"])
    assert torch.equal(aux["sort_idx"].cpu().long(), ref.aux["sort_idx"])
    assert scaled_err(depth.cpu().numpy(), ref.depth.numpy()) < TOL
    assert scaled_err(pixels.cpu().numpy(), ref.pixels.numpy()) < TOL


def _oracle_case(dev, variant, B, R, S, V, H, clamp="relu", noise=0.0, white=True, last=False, seed=0, check_grads=False,
                 precision="fp32", Z=32, head_scale=30.0):
    """Random inputs of an arbitrary shape: HIP vs the CPU oracle on the same rays / draws, fine depths forced.
    `precision` may be a tuple: every listed forward kernel is checked against the one oracle run."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    from oracle import render_oracle as O, checks as K
    torch.manual_seed(seed)
    np.random.seed(seed)
    has_glob = O.FIELD_SPECS[variant].has_global
    gen = ImplicitGenerator3d(variant, Z if has_glob else 32, 32, 4, H)
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= head_scale
    fvol, glob = torch.randn(B, 32, V, V, V) * 0.5, (torch.randn(B, Z) if has_glob else None)
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y")
    P = R * R
    rng = {"u_strat": torch.rand(B, P, S), "eps_coarse": torch.randn(B, P, S), "u_fine": torch.rand(B, P, S),
           "eps_final": torch.randn(B, P, 2 * S)}
    params = {k: v.detach() for k, v in gen.siren.state_dict().items()}
    with torch.no_grad():
        ref = O.render(variant, params, fvol, glob, cam, R, 49.13, 0.25, 1.95, S, True, clamp, noise, white, last,
                       rng["u_strat"], rng["eps_coarse"], rng["u_fine"], rng["eps_final"])
    gen.to(dev)
    gen.set_device(dev)
    r = {k: v.to(dev) for k, v in rng.items()}
    r["fine_z"] = ref.aux["fine_z"].to(dev)
    z = (fvol.to(dev), glob.to(dev)) if has_glob else fvol.to(dev)
    # the field in float64 at (a strided sample of at most ~1 M of) the coarse sample positions: the yardstick of accuracy_vs_fp64
    npi = P * S
    sub = torch.arange(0, npi, max(1, (B * npi + (1 << 20) - 1) >> 20))
    exact = K.field_fp64(variant, params, fvol, glob, ref.aux["coarse_points"].reshape(B, npi, 3)[:, sub])
    out = {}
    for prec in ((precision,) if isinstance(precision, str) else precision):
        gen.siren.precision = prec
        aux = {}
        with torch.no_grad():
            px, dp = gen(z, cam.to(dev), R, 49.13, 0.25, 1.95, S, True, clamp_mode=clamp, nerf_noise=noise, white_back=white,
                         last_back=last, _rng=r, _aux=aux)
        assert torch.equal(aux["coarse_points"].cpu(), ref.aux["coarse_points"]), prec
        assert torch.equal(aux["fine_points"].cpu(), ref.aux["fine_points"]), prec
        assert merge_order_matches(aux["sort_idx"].cpu().numpy(), ref.aux["sort_idx"].numpy(), ref.aux["fine_z"].numpy(),
                                   ref.aux["coarse_z"].numpy()), prec
        e_c, e_f = rgb_sigma_err(aux["coarse_rgb_sigma"], ref.aux["coarse_rgb_sigma"]), rgb_sigma_err(aux["fine_rgb_sigma"], ref.aux["fine_rgb_sigma"])
        assert e_c < TOL and e_f < TOL, (prec, e_c, e_f)
        same = (aux["inds"].cpu() == ref.aux["inds"]).float().mean().item()
        assert same > 0.99, (prec, same)
        # "as accurate as the reference" measured: both fp32 results against the field in float64 at the same positions
        acc = K.accuracy_vs_fp64(aux["coarse_rgb_sigma"].cpu().reshape(B, -1, 4)[:, sub], ref.aux["coarse_rgb_sigma"].reshape(B, -1, 4)[:, sub], exact)
        if B * npi >= 4096:      # (a maximum over a handful of points is no statistic: recorded, not asserted)
            assert acc["hip_vs_fp64"] <= 2 * acc["ref_vs_fp64"] + 1e-6, (prec, acc)
        if P > 1:      # (a single ray has no rms to scale by)
            # rays the reference itself is discontinuous on (last-sample density at a zero crossing under relu): the HIP value must
            # equal ONE of the reference algorithm's two branches there -- a positive check, nothing is left out of the maximum
            edge = K.knife_edge_rays(ref.aux, clamp)
            branches = None
            if edge is not None and edge.any():
                assert edge.float().mean().item() < 1e-3, edge.float().mean().item()
                branches = K.knife_edge_branches(ref.aux, R, 49.13, noise, white, last, rng["eps_final"])
            e_p, e_d, n_edge = K.image_err_with_knife_edges(px, dp, ref.pixels, ref.depth, edge, branches)
            assert e_p < 2 * TOL and e_d < 2 * TOL, (prec, e_p, e_d, n_edge)
        else:
            e_p = e_d = np.abs(px.cpu().numpy() - ref.pixels.numpy()).max()
            n_edge = 0
            assert e_p < 2e-4
        out[prec] = dict(coarse=e_c, fine=e_f, inds_same=same, pixels=e_p, depth=e_d, knife_edge_rays=n_edge, **acc,
                         survey_rs=survey_metric_pass(aux["coarse_rgb_sigma"].cpu().numpy(), ref.aux["coarse_rgb_sigma"].numpy()),
                         survey_px=survey_metric_pass(px.cpu().numpy(), ref.pixels.numpy()))
    return out


def rgb_sigma_err(a, b):
    """scaled_err of colour and of density, each against its own scale (they are different quantities: rgb is O(1), the
    density head is scaled up to O(10) in these tests): the larger of the two."""
    a = a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return max(scaled_err(a[..., :3], b[..., :3]), scaled_err(a[..., 3], b[..., 3]))


def merge_order_matches(sort_idx, ref_sort_idx, fine_z, coarse_z):
    """The merge permutation equals the reference's wherever the merged depths are distinct, and the sorted depths are
    identical bit for bit everywhere.  With 2S = 128 samples per ray a handful of rays per image hold two samples of EQUAL
    fp32 depth (birthday odds ~5e-4 per ray at 128 samples); torch.sort is not a stable sort, so for such twins -- and only
    for them -- either order is the reference's answer (the compositing sees the sorted depths, which agree)."""
    si, sr = np.asarray(sort_idx).astype(np.int64), np.asarray(ref_sort_idx).astype(np.int64)
    allz = np.concatenate([fine_z, coarse_z], -1)
    za, zr = np.take_along_axis(allz, si, -1), np.take_along_axis(allz, sr, -1)
    tie = np.zeros(sr.shape, bool)
    tie[..., 1:] |= zr[..., 1:] == zr[..., :-1]
    tie[..., :-1] |= zr[..., :-1] == zr[..., 1:]
    return bool(np.array_equal(za, zr) and not ((si != sr) & ~tie).any())


def survey_metric_pass(a, b):
    """Fraction of elements inside the gate SURVEY.md 8(d) wrote down, |a-b| <= 1e-4 * max(|b|, 1e-3), reported next to
    scaled_err (whose floor is rms(b)): see DESIGN.md section 4 for why the gate of the tests is the latter."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= 1e-4 * np.maximum(np.abs(b), 1e-3)))


def test_benchmarked_shape_matches_oracle(dev):
    """The shape bench.py times -- SHORTSIREN_FG, hidden 256, 64^3 volume, 128x128 rays x (64 + 64) samples -- with TWO
    images (the 8 XCD bands run over both, the fp16x3 kernel restages the FiLM vectors at the image switch, two 32-point tiles
    per ray), against the CPU oracle on identical inputs and draws, both forward kernels: sample positions and merge order
    bit-exact, rgb / sigma of both passes within 1e-4, image within 2e-4 (fine depths forced, as everywhere)."""
    res = _oracle_case(dev, "SHORTSIREN_FG", B=2, R=128, S=64, V=64, H=256, Z=256, precision=("fp32", "fp16x3"), seed=11)
    print("benchmarked shape:", res)
    # margin guard: the gate is 1e-4; a kernel change that eats the rest of the margin at the benchmarked shape fails HERE, loudly,
    # before it can flip a fixture (round 2 spent the margin from 2e-5 to 7e-5 on the folded epilogue: VERDICT r02 weak #1)
    for prec, r in res.items():
        assert max(r["coarse"], r["fine"]) < 0.9e-4, (prec, r)


def test_config5_shape_matches_oracle(dev):
    """BASELINE config 5's shape: 256x256 rays x (96 + 96) samples (three 32-point tiles per ray, 12.6 M field evaluations per
    image), 64^3 volume, hidden 256, one image, against the CPU oracle on identical inputs and draws -- the exact fp32 kernel and
    the fp16x3 split at the 1e-4 gate (the single-pass fp16 arithmetic that config names is covered, at its own tolerance, by
    test_single_pass_fp16 and at this size by test_full_size_properties)."""
    res = _oracle_case(dev, "SHORTSIREN_FG", B=1, R=256, S=96, V=64, H=256, Z=256, precision=("fp32", "fp16x3"), seed=13)
    print("config-5 shape:", res)


def test_config2_shape_matches_oracle(dev):
    """BASELINE config 2 as it is written: 64x64 rays x 24 samples, 64^3 volume, hidden 256, one image (the `short_fg_64x24`
    fixture of the reference uses a 24^3 volume to stay small)."""
    res = _oracle_case(dev, "SHORTSIREN_FG", B=1, R=64, S=24, V=64, H=256, Z=256, clamp="softplus", noise=0.5,
                       precision=("fp32", "fp16x3"), seed=12)
    print("config-2 shape:", res)


@pytest.mark.parametrize("shape", [
    dict(B=3, R=5, S=7, V=9),          # ragged: 175 points per image -> padded last tile, odd everything
    dict(B=1, R=1, S=2, V=2),          # minimum sizes
    dict(B=1, R=3, S=128, V=6),        # maximum samples per ray (four 64-lane chunks after the merge)
    dict(B=2, R=6, S=33, V=5),         # samples straddle the 32-point tiles and the 64-lane chunks
    dict(B=1, R=7, S=64, V=33),        # bench-like S, odd volume
])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_ragged_and_extreme_shapes(dev, shape, precision):
    _oracle_case(dev, "SHORTSIREN_FG", H=64, precision=precision, **shape)


@pytest.mark.parametrize("H", [128, 256])
def test_split_precision_wide_ragged(dev, H):
    """fp16x3 at the widths whose kernels stage weight units differently (4 / 8 output tiles), three images of a ragged
    size (idle waves in the last tile group of every image, FiLM vectors restaged per image), softplus + noise."""
    _oracle_case(dev, "SHORTSIREN_FG", B=3, R=9, S=11, V=12, H=H, clamp="softplus", noise=0.3, precision="fp16x3", seed=H)
    _oracle_case(dev, "TALLSIREN_FG", B=2, R=6, S=13, V=10, H=H, precision="fp16x3", seed=H + 1)


def test_extreme_shapes_other_families(dev):
    _oracle_case(dev, "TALLSIREN_dRes", B=2, R=5, S=9, V=7, H=64, clamp="softplus", noise=0.4, last=True)
    # the deepest residual chain at the width where the fp32 residual kernel spills registers (H = 256), ragged tiles
    _oracle_case(dev, "TALLSIREN_dResLong", B=2, R=7, S=19, V=9, H=256, precision=("fp32", "fp16x3"), head_scale=10.0, seed=5)
    _oracle_case(dev, "TALLSIREN_dResLong", B=1, R=5, S=9, V=7, H=128, clamp="softplus", noise=0.4, last=True, head_scale=10.0)
    _oracle_case(dev, "DOUBLESIREN_FG", B=1, R=4, S=17, V=4, H=128, white=False)


@pytest.mark.parametrize("shape", [(2, 1000, 256, 256), (1, 37, 64, 32), (3, 5003, 128, 64), (2, 4099, 256, 192), (1, 70000, 256, 96)])
def test_weight_grad_kernel(dev, shape):
    """cnerf_weight_grad (per-image G^T X and column sums over a chunk) against float64 matmuls: ragged point counts, every
    supported width, both outputs accumulate."""
    import cnerf_amd
    from cnerf_amd import ops, _lib as L
    cnt, npi, H, K = shape
    torch.manual_seed(cnt * 7 + K)
    G, X = torch.randn(cnt, npi, H, device=dev), torch.randn(cnt, npi, K, device=dev)
    dW, cs = torch.ones(cnt, H, K, device=dev), torch.ones(cnt, H, device=dev)          # accumulated into
    L.check(L.lib().cnerf_weight_grad(cnt, npi, H, K, L.ptr(G), L.ptr(X), L.ptr(dW), L.ptr(cs), ops._stream()), "weight_grad")
    ref = torch.bmm(G.transpose(1, 2).double(), X.double()) + 1
    rcs = G.double().sum(1) + 1
    assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 5e-6
    assert ((cs.double() - rcs).abs().max() / rcs.abs().max()).item() < 5e-6


def test_precisions_agree_on_random_shapes(dev):
    """The two forward kernels against each other on two dozen random (family, width, batch, image size, samples, volume)
    combinations, full hierarchical render with the fine depths of the fp32 run forced into the fp16x3 run: pixels within
    2e-4, merge order identical.  Catches indexing slips (tile groups, idle waves, image switches, unit sequence lengths)
    that a fixed fixture shape can hide."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from oracle import render_oracle as O
    rs = np.random.RandomState(7)
    fams = ["SHORTSIREN_FG", "DOUBLESIREN_FG", "TALLSIREN_FG", "SingleSIREN_dg", "SHORTSIREN_F", "SHORTSIREN_FRes", "TALLSIREN_dRes",
            "TALLSIREN_dResLong"]
    for case in range(24):
        variant = fams[case % len(fams)]
        H = int(rs.choice([64, 128, 256]))
        B, R, S, V = int(rs.randint(1, 4)), int(rs.randint(1, 14)), int(rs.randint(2, 70)), int(rs.randint(2, 20))
        torch.manual_seed(case)
        has_glob = O.FIELD_SPECS[variant].has_global
        gen = ImplicitGenerator3d(variant, 32, 32, 4, H).to(dev)
        gen.set_device(dev)
        with torch.no_grad():
            gen.siren.final_layer.weight[3] *= 20
        fvol = torch.randn(B, 32, V, V, V, device=dev) * 0.5
        z = (fvol, torch.randn(B, 32, device=dev)) if has_glob else fvol
        cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
        cam[:, 2, 3] = -1.0
        cam[:, 0, 3] = torch.linspace(-0.2, 0.2, B, device=dev)
        rng = {"u_strat": torch.rand(B, R * R, S, device=dev), "u_fine": torch.rand(B, R * R, S, device=dev),
               "eps_coarse": torch.randn(B, R * R, S, device=dev), "eps_final": torch.randn(B, R * R, 2 * S, device=dev)}
        outs = {}
        for prec in ("fp32", "fp16x3"):
            gen.siren.precision = prec
            aux = {}
            r = dict(rng)
            if prec == "fp16x3":
                r["fine_z"] = outs["fp32"][2]["fine_z"]
            with torch.no_grad():
                px, dp = gen(z, cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus", nerf_noise=0.2, white_back=True, _rng=r, _aux=aux)
            outs[prec] = (px, dp, aux)
        a, b = outs["fp32"], outs["fp16x3"]
        tag = f"case {case}: {variant} H={H} B={B} R={R} S={S} V={V}"
        assert torch.equal(a[2]["coarse_z"], b[2]["coarse_z"]), tag
        assert torch.equal(a[2]["sort_idx"], b[2]["sort_idx"]), tag
        assert (a[0] - b[0]).abs().max().item() < 2e-4, tag
        assert (a[1] - b[1]).abs().max().item() < 2e-4, tag


def test_channels_last_volume_is_taken_without_a_copy(dev):
    """A feature volume in torch.channels_last_3d memory format renders the same image as its contiguous twin, and the
    channel-last view handed to the kernels aliases the caller's storage."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(5)
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, 64).to(dev)
    gen.set_device(dev)
    fvol = torch.randn(2, 32, 9, 9, 9, device=dev)
    fcl = fvol.to(memory_format=torch.channels_last_3d)
    assert ops.channel_last(fcl).data_ptr() == fcl.data_ptr()
    glob = torch.randn(2, 32, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(2, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(2, 36, 8, device=dev), "u_fine": torch.rand(2, 36, 8, device=dev)}
    with torch.no_grad():
        a = gen((fvol, glob), cam, 6, 49.13, 0.25, 1.95, 8, True, clamp_mode="relu", nerf_noise=0.0, _rng=rng)
        b = gen((fcl, glob), cam, 6, 49.13, 0.25, 1.95, 8, True, clamp_mode="relu", nerf_noise=0.0, _rng=rng)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # and gradients flow back to a channels_last leaf
    leaf = fcl.clone().requires_grad_(True)
    gen.train()
    px, dp = gen((leaf, glob), cam, 6, 49.13, 0.25, 1.95, 8, True, clamp_mode="relu", nerf_noise=0.0, _rng=rng)
    (px.square().mean() + dp.mean()).backward()
    assert leaf.grad is not None and torch.isfinite(leaf.grad).all() and leaf.grad.abs().sum() > 0


def test_extract_shapes_grid(dev):
    """Density grid helper (second consumer of the siren sub-API, extract_shapes.py:41-78): chunked evaluation equals one
    direct field call, the sample lattice spans the cube."""
    import cnerf_amd
    from cnerf_amd import extract_shapes as X
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(11)
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, 64).to(dev)
    gen.set_device(dev)
    z = (torch.randn(1, 32, 8, 8, 8, device=dev), torch.randn(1, 32, device=dev))
    N = 12
    pts, corner, pitch = X.create_samples(N, cube_length=1.2)
    assert pts.shape == (1, N ** 3, 3) and abs(float(pts.min()) + 0.6) < 1e-6 and abs(float(pts[..., 2].max()) - 0.6) < 1e-5
    grid = X.sample_generator(gen, z, voxel_resolution=N, max_points=500)        # ragged chunks on purpose
    with torch.no_grad():
        direct = gen.siren(pts.to(dev), z, N, 1)[..., 3].reshape(N, N, N).cpu().numpy()
    assert grid.shape == (N, N, N) and np.array_equal(grid, direct)


def test_scatter_is_the_adjoint_of_gather(dev):
    """cnerf_scatter_features against cnerf_gather_features: <gather(v), g> == <v, scatter(g)> for random v, g, points (incl.
    points outside the volume, which clamp to the border like the lookup does)."""
    import ctypes as C
    import cnerf_amd
    from cnerf_amd import ops, _lib as L
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(3)
    net = ImplicitGenerator3d("SHORTSIREN_FG", 16, 32, 4, 64).to(dev).siren
    B, V, n = 2, 6, 1000
    vol = torch.randn(B, V, V, V, 32, device=dev)
    pts = (torch.rand(B, n, 3, device=dev) - 0.5) * 1.6            # |coordinate| up to 0.8 > 0.6: border clamp exercised
    g = torch.randn(B, n, 32, device=dev)
    feat = ops.gather_features(net, vol, pts)
    out = torch.zeros_like(vol)
    cfg = ops.make_cfg(net, B, V)
    L.check(L.lib().cnerf_scatter_features(C.byref(cfg), L.ptr(pts), n, L.ptr(g), L.ptr(out), ops._stream()), "scatter")
    lhs, rhs = (feat.double() * g.double()).sum().item(), (vol.double() * out.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_bad_arguments_are_refused(dev):
    """Shapes outside the supported range come back as CnerfError with a message, never as a wrong image."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 16, 32, 4, 64).to(dev)
    gen.set_device(dev)
    fv, gl, cam = torch.zeros(1, 32, 4, 4, 4, device=dev), torch.zeros(1, 16, device=dev), torch.eye(4, device=dev)[None]
    with pytest.raises(cnerf_amd._lib.CnerfError, match="S="):
        gen((fv, gl), cam, 4, 30.0, 0.1, 1.0, 129, True, clamp_mode="relu", nerf_noise=0.0)
    with pytest.raises(cnerf_amd._lib.CnerfError, match="S="):
        gen((fv, gl), cam, 4, 30.0, 0.1, 1.0, 1, True, clamp_mode="relu", nerf_noise=0.0)
    with pytest.raises(TypeError):
        gen((fv, gl), cam, 4, 30.0, 0.1, 1.0, 8, True, clamp_mode=None, nerf_noise=0.0)
    with pytest.raises(cnerf_amd._lib.CnerfError):
        gen((torch.zeros(1, 16, 4, 4, 4, device=dev), gl), cam, 4, 30.0, 0.1, 1.0, 8, True, clamp_mode="relu", nerf_noise=0.0)
    with pytest.raises(KeyError):            # the reference reads kwargs["clamp_mode"] unconditionally too
        gen((fv, gl), cam, 4, 30.0, 0.1, 1.0, 8, True, nerf_noise=0.0)


@pytest.mark.parametrize("size", [(128, 64, "fp32"), (128, 64, "fp16x3"), (256, 96, "fp32"), (256, 96, "fp16x3"), (256, 96, "fp16")])
def test_full_size_properties(dev, size):
    """BASELINE sizes 128x128x64 (configs 3/4) and 256x256x96 (config 5), B=1: properties that need no oracle -- weights
    form a sub-probability, white background fills the missing mass, depth within [ray_start, ray_end]*dir_z, determinism
    across two runs; every forward precision (config 5 also with the single-pass fp16 kernel it names)."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(0)
    R, S, prec = size
    V = 64
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
    gen.set_device(dev)
    gen.siren.precision = prec
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 40
    fvol, glob = torch.randn(1, 32, V, V, V, device=dev), torch.randn(1, 256, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).clone()
    cam[0, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(1, R * R, S, device=dev), "u_fine": torch.rand(1, R * R, S, device=dev)}
    outs = []
    for _ in range(2):
        aux = {}
        with torch.no_grad():
            px, dp = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True,
                         _rng=rng, _aux=aux)
        outs.append((px.clone(), dp.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    w = aux["final_weights"]
    assert torch.isfinite(px).all() and torch.isfinite(dp).all()
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    assert (px <= 1 + 1e-5).all() and (px >= -1 - 1e-5).all()
    assert (dp >= 0).all() and (dp <= 1.95 + 1e-4).all()
    z = aux["fine_z"]
    assert (z >= 0.25 - 0.02).all() and (z <= 1.95 + 0.02).all()
    srt = torch.gather(torch.cat([aux["fine_z"], aux["coarse_z"]], -1), -1, aux["sort_idx"].long())
    assert (srt[..., 1:] >= srt[..., :-1]).all()           # merged depths are sorted
    assert (aux["sort_idx"].sort(-1)[0] == torch.arange(2 * S, device=dev)).all()   # and a permutation


def to_tb16(m):
    """(cnt, npi, Ch) -> TB16 fp16 buffer (cnt * tiles, Ch / 32, 32, 32), rows past the end of an image zero (bwd16.hpp)."""
    cnt, npi, ch = m.shape
    tiles = (npi + 31) // 32
    pad = torch.zeros(cnt, tiles * 32, ch, dtype=torch.float16, device=m.device)
    pad[:, :npi] = m.half()
    return pad.reshape(cnt * tiles, 32, ch // 32, 32).permute(0, 2, 1, 3).contiguous(), tiles


@pytest.mark.parametrize("shape", [(2, 1000, 256, 256), (1, 37, 64, 32), (3, 5003, 128, 64), (2, 4099, 256, 192), (1, 70000, 256, 96),
                                   (2, 3000, 4, 256), (1, 999, 4, 64)])
def test_weight_grad16_kernel(dev, shape):
    """cnerf_weight_grad16 (fp16 MFMA over TB16 buffers, transposed LDS reads) against float64 matmuls of the SAME fp16 values:
    ragged point counts, every width, 4-row head case, column runs of 8 / 4 / 2 / 1 tiles, scale undone, outputs accumulate."""
    import cnerf_amd
    from cnerf_amd import ops, _lib as L
    cnt, npi, H, K = shape
    torch.manual_seed(cnt * 7 + K)
    G, X = torch.randn(cnt, npi, max(H, 32), device=dev), torch.randn(cnt, npi, K, device=dev)
    if H < 32:
        G[..., H:] = 0
    scale = 8.0
    g16, tiles = to_tb16(G * scale)
    x16, _ = to_tb16(X)
    dW, cs = torch.ones(cnt, H, K, device=dev), torch.ones(cnt, H, device=dev)          # accumulated into
    inv = torch.tensor([1.0 / scale], device=dev)
    L.check(L.lib().cnerf_weight_grad16(cnt, tiles, H, g16.shape[1], x16.shape[1], L.ptr(g16), L.ptr(x16), L.ptr(dW), L.ptr(cs), L.ptr(inv),
                                        ops._stream()), "weight_grad16")
    Gh, Xh = (G * scale).half().double()[..., :H] / scale, X.half().double()
    ref = torch.bmm(Gh.transpose(1, 2), Xh) + 1
    rcs = Gh.sum(1) + 1
    assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 5e-6
    assert ((cs.double() - rcs).abs().max() / rcs.abs().max()).item() < 5e-6


@pytest.mark.parametrize("backward_precision", ["fp32", "fp16"])
@pytest.mark.parametrize("shape", [dict(B=2, R=5, S=7, V=9, H=64), dict(B=1, R=3, S=33, V=6, H=128), dict(B=3, R=4, S=9, V=5, H=256)])
def test_backward_ragged_shapes_vs_oracle_autograd(dev, shape, backward_precision):
    """Gradients at point counts that are NOT a multiple of the 32-point tile (175, 297, 144 points per image: padded last
    tiles, idle waves in the last tile group, several images) against autograd through the CPU oracle, fine depths forced:
    every parameter, the FiLM mapping, the feature volume and the global feature."""
    _ragged_backward_case(dev, shape, "SHORTSIREN_FG", backward_precision)


@pytest.mark.parametrize("shape", [dict(B=2, R=5, S=7, V=9, H=64), dict(B=1, R=3, S=33, V=6, H=128), dict(B=3, R=4, S=9, V=5, H=256)])
def test_backward_per_point_film_vs_oracle_autograd(dev, shape):
    """The same for TALLSIREN (per-point FiLM, siren.py:232-331): storing forward + gradient chain kernels, weight-gradient
    reductions and the mapping network's GEMMs against autograd through the CPU oracle, at all three widths and ragged tiles --
    the exact fp32 path (field_pw_backward_kernel + library GEMMs) and the half-precision one (field_pw16 storing forward,
    pw_deriv_kernel, chain_pre_kernel, pw_gm_kernel, weight_grad16 behind cnerf_render_backward)."""
    _ragged_backward_case(dev, shape, "TALLSIREN", ("fp32", "fp16"))


@pytest.mark.parametrize("variant", ["TALLSIREN_dRes", "SHORTSIREN_FRes", "TALLSIREN_dResLong"])
@pytest.mark.parametrize("shape", [dict(B=2, R=5, S=7, V=9, H=64), dict(B=1, R=4, S=33, V=6, H=256)])
def test_backward_residual_blocks_vs_oracle_autograd(dev, shape, variant):
    """Residual-block networks (sin(x + W2 sin(W1 x + b1) + b2), siren.py:218-230; two, one and four blocks) through both
    backward paths against autograd through the CPU oracle: the identity path of the fp32 chain (re-read rows) and of the
    half-precision chain (operand fragments kept in registers across two products), ragged tiles, narrow and wide."""
    _ragged_backward_case(dev, shape, variant, ("fp32", "fp16"))


@pytest.mark.parametrize("variant", ["SHORTSIREN_FRes", "TALLSIREN_FG", "TALLSIREN"])
def test_backward_with_dropout_vs_oracle_autograd(dev, variant):
    """Training mode with drop_out > 0 at hidden 256 and ragged tiles, decisions injected on both sides: a network whose dropout
    layers are interleaved with residual blocks (which have none: the layer counter of the decisions must skip them), the
    eight-layer FiLM network and the per-point FiLM family, forward re-run and gradient chain against autograd through the oracle."""
    _ragged_backward_case(dev, dict(B=2, R=4, S=9, V=6, H=256), variant, "fp32", drop_p=0.3)


def test_benchmarked_shape_backward_vs_oracle_autograd(dev):
    """BASELINE configs 3/4 train at this shape: gradients of one image at 128x128 rays x (64 + 64) samples, 64^3 volume, hidden
    256 (two 32-point tiles per ray, 32 k tiles per pass, the LDS-resident head) against autograd through the CPU oracle on the
    same inputs and draws, fine depths forced -- the exact fp32 backward and the half-precision one against the same oracle run
    (fp32 and float64: ~1.5 min of host time, ~60 GB of host memory)."""
    _ragged_backward_case(dev, dict(B=1, R=128, S=64, V=64, H=256), "SHORTSIREN_FG", ("fp32", "fp16"))


def _ragged_backward_case(dev, shape, variant, backward_precisions, drop_p=0.0):
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    from oracle import render_oracle as O
    B, R, S, V, H = (shape[k] for k in "BRSVH")
    torch.manual_seed(B * 100 + S)
    np.random.seed(B * 100 + S)
    Z = 32
    from cnerf_amd.generators.siren import FIELD_SPECS
    has_glob = variant != "TALLSIREN" and FIELD_SPECS[variant].has_global
    gen = (ImplicitGenerator3d(variant, Z, 32, 4, H, drop_out=drop_p) if has_glob else
           ImplicitGenerator3d(variant, 32, 3 if variant == "TALLSIREN" else 32, 4, H, drop_out=drop_p))
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 20
    fvol, glob = torch.randn(B, 32, V, V, V) * 0.5, (torch.randn(B, Z) if has_glob else None)
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y")
    P = R * R
    rng = {"u_strat": torch.rand(B, P, S), "eps_coarse": torch.randn(B, P, S), "u_fine": torch.rand(B, P, S), "eps_final": torch.randn(B, P, 2 * S)}
    drop = {}
    if drop_p:          # training mode: keep decisions of the two field passes, one (B, P*S, H) slab per FiLM / sine layer (not per residual block)
        n_drop = sum(1 for k in gen.siren.spec.layers if k != "res")
        drop = {k: (torch.rand(n_drop, B, P * S, H) >= drop_p).to(torch.uint8) for k in ("drop_coarse", "drop_fine")}
    def oracle_grads(dtype):
        c = lambda t: t.detach().clone().to(dtype)
        params = {k: c(v).requires_grad_(True) for k, v in gen.siren.state_dict().items()}
        fv_r, gl_r = c(fvol).requires_grad_(True), (c(glob).requires_grad_(True) if has_glob else None)
        torch.set_default_dtype(dtype)
        try:
            ref = O.render(variant, params, fv_r, gl_r, c(cam), R, 49.13, 0.25, 1.95, S, True, "softplus", 0.3, True, False,
                           c(rng["u_strat"]), c(rng["eps_coarse"]), c(rng["u_fine"]), c(rng["eps_final"]),
                           forced_fine_z=None if dtype == torch.float32 else forced, drop_p=drop_p, **drop)
        finally:
            torch.set_default_dtype(torch.float32)
        leaves = [fv_r] + ([gl_r] if has_glob else []) + list(params.values())
        grads = torch.autograd.grad(ref.pixels.square().mean() + ref.depth.mean(), leaves)
        names = ["feature_volume"] + (["global_feature"] if has_glob else []) + list(params.keys())
        return ref, {k: v.float().numpy() for k, v in zip(names, grads)}

    forced = None
    ref, want = oracle_grads(torch.float32)
    forced = ref.aux["fine_z"].detach().double()
    _, exact = oracle_grads(torch.float64)                 # same sample positions, exact arithmetic: the reference's own noise floor
    gen.to(dev)
    gen.set_device(dev)
    gen.train()
    r = {k: v.to(dev) for k, v in {**rng, **drop}.items()}
    r["fine_z"] = ref.aux["fine_z"].detach().to(dev)
    for backward_precision in ([backward_precisions] if isinstance(backward_precisions, str) else backward_precisions):
        gen.siren.precision = "fp32" if backward_precision == "fp32" else "fp16x3"
        gen.siren.backward_precision = backward_precision
        gen.zero_grad()
        fv, gl = fvol.to(dev).requires_grad_(True), (glob.to(dev).requires_grad_(True) if has_glob else None)
        px, dp = gen((fv, gl) if has_glob else fv, cam.to(dev), R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus", nerf_noise=0.3,
                     white_back=True, _rng=r)
        (px.square().mean() + dp.mean()).backward()
        got = {"feature_volume": fv.grad}
        if has_glob:
            got["global_feature"] = gl.grad
        got.update({k: p.grad for k, p in gen.siren.named_parameters()})
        for k, w in want.items():
            floor = scaled_err(w, exact[k])
            e, l2 = scaled_err(got[k].cpu().numpy(), w), rel_l2(got[k].cpu().numpy(), w)
            if backward_precision == "fp32":
                assert e < max(2e-3, 2.5 * floor), (backward_precision, k, e, floor)
            else:
                assert l2 < max(2e-3, 2.5 * rel_l2(w, exact[k])) and e < max(5e-2, 2.5 * floor), (backward_precision, k, e, l2, floor)


def test_fancy_integration_fill_modes(dev):
    """The debug paints of fancy_integration (volumetric_rendering.py:62-67) against the oracle's composite + the reference's
    two formulas: "debug" paints rays whose weights sum below 0.9 red, "weight" returns the weight sum as colour."""
    from cnerf_amd.generators.volumetric_rendering import fancy_integration
    from oracle import render_oracle as O
    torch.manual_seed(2)
    rs = torch.randn(2, 50, 9, 4)
    rs[..., 3] *= 3
    z = torch.sort(torch.rand(2, 50, 9) * 1.5 + 0.25, -1)[0]
    for last_back in (False, True):
        rgb_ref, _, w = O.composite(rs, z, None, 0.0, "relu", white_back=True, last_back=False)
        wsum = w.sum(-1, keepdim=True)
        rgb0, _, _ = O.composite(rs, z, None, 0.0, "relu", white_back=True, last_back=last_back)
        for mode in ("debug", "weight"):
            got, _, _ = fancy_integration(rs.to(dev), z.unsqueeze(-1).to(dev), dev, noise_std=0.0, last_back=last_back, white_back=True,
                                          clamp_mode="relu", fill_mode=mode)
            want = torch.where(wsum < 0.9, torch.tensor([1.0, 0.0, 0.0]).expand_as(rgb0), rgb0) if mode == "debug" else wsum.expand_as(rgb0)
            near = (wsum - 0.9).abs() < 1e-5            # rays sitting on the threshold may fall either way
            assert ((got.cpu() - want).abs().max(-1)[0][~near.squeeze(-1)] < 1e-5).all(), (mode, last_back)


def test_philox_draws_match_numpy_twin(dev):
    """cnerf_philox_fill (the draws the kernels generate under cnerf_cfg.philox) against oracle/philox.py, itself pinned by the
    Philox4x32-10 known-answer vectors: uniforms bit for bit, normals to 1e-6 (device logf / cosf vs NumPy's)."""
    from cnerf_amd import ops
    from oracle import philox as P
    seed, off, n = 0x1234567890ABCDEF, 41, 70001
    for stream in range(4):
        u = ops.philox_fill(seed, off, stream, n, False, dev).cpu().numpy()
        assert np.array_equal(u, P.uniform(seed, off, stream, n)), stream
        g = ops.philox_fill(seed, off, stream, n, True, dev).cpu().numpy()
        assert np.abs(g - P.normal(seed, off, stream, n)).max() < 2e-6, stream
    assert not np.array_equal(ops.philox_fill(seed, off + 1, 0, 100, False, dev).cpu().numpy(), P.uniform(seed, off, 0, 100))


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_in_kernel_philox_equals_injected_draws(dev, precision):
    """A render whose four random draws are generated inside the kernels (cnerf_cfg.philox: no tensors) is bit-identical to the
    render with the same draws injected as tensors (cnerf_philox_fill), forward and -- for the gradients -- backward: stratified
    jitter in the field kernels, density noise and inverse-CDF draws in the per-ray kernels, the re-computed positions of the
    backward.  Ragged size, several images, noise on."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(9)
    B, R, S, V, H = 3, 7, 11, 9, 64
    P_ = R * R
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, H).to(dev)
    gen.set_device(dev)
    gen.siren.precision = precision
    gen.train()
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 20
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev), torch.randn(B, 32, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    cam[:, 0, 3] = torch.linspace(-0.2, 0.2, B, device=dev)
    seed, off = 987654321, 5
    inj = {"u_strat": ops.philox_fill(seed, off, 0, B * P_ * S, False, dev), "eps_coarse": ops.philox_fill(seed, off, 1, B * P_ * S, True, dev),
           "u_fine": ops.philox_fill(seed, off, 2, B * P_ * S, False, dev), "eps_final": ops.philox_fill(seed, off, 3, B * P_ * 2 * S, True, dev)}
    outs = []
    for rng in ({"philox": (seed, off)}, inj):
        fv, gl = fvol.clone().requires_grad_(True), glob.clone().requires_grad_(True)
        gen.zero_grad()
        aux = {}
        px, dp = gen((fv, gl), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus", nerf_noise=0.4, white_back=True, _rng=rng, _aux=aux)
        (px.square().mean() + dp.mean()).backward()
        outs.append((px.detach().clone(), dp.detach().clone(), aux["coarse_z"].clone(), aux["fine_z"].clone(), fv.grad.clone(), gl.grad.clone(),
                     gen.siren.network[0].layer.weight.grad.clone()))
    a, b = outs
    for i in range(4):
        assert torch.equal(a[i], b[i]), i                       # pixels, depth, jittered depths, resampled depths: bit for bit
    for i in range(4, 7):                                        # gradients: the volume scatter's atomics reorder sums at the 1e-7 level
        assert (a[i] - b[i]).abs().max().item() <= 1e-5 * b[i].abs().max().item(), i
    # and the module-level switch: two forwards draw different numbers, a re-seeded twin the same ones
    gen.eval()
    gen.rng_mode = "philox"
    torch.cuda.manual_seed(77)
    with torch.no_grad():
        p1, _ = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
        p2, _ = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
    assert not torch.equal(p1, p2)
    gen._rng_calls -= 2
    with torch.no_grad():
        p3, _ = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
    assert torch.equal(p1, p3)
    # the counter follows generator.step (what a checkpoint stores): a resumed run continues the stream, it does not replay step 0's
    # draws; and a second generator under the same seed (a frozen teacher next to its student) draws a stream of its own
    gen.step = 7
    with torch.no_grad():
        p4, _ = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
    assert not torch.equal(p4, p1) and not torch.equal(p4, p2)
    import copy
    twin = copy.deepcopy(gen)
    twin._rng_salt = gen._rng_salt + 1            # what a second ImplicitGenerator3d(...) of the process gets
    gen.step, twin.step = 0, 0
    with torch.no_grad():
        q1, _ = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
        q2, _ = twin((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0)
    assert torch.equal(q1, p1) and not torch.equal(q2, q1)


def _dropout_rng(g, dev):
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final", "drop_coarse", "drop_fine") if g.get(k) is not None}
    if g.meta["hierarchical"]:
        rng["fine_z"] = G(g["fine_z"], dev)
    return rng


@pytest.mark.parametrize("name", DROP_GOLDEN)
def test_dropout_training_mode_matches_reference(golden, dev, name):
    """The reference in training mode with drop_out > 0 (siren.py:158-159,175-176,197-198), its own F.dropout keep decisions
    injected (cnerf_rng.drop_coarse / drop_fine), fine depths forced: field outputs, image, depth at the 1e-4 gate and every
    gradient at the tolerance of test_backward_teacher_forced -- FiLM, per-point FiLM (TALLSIREN) and plain-sine networks."""
    g = golden(name)
    m = g.meta
    gen = make_generator(g, dev)
    gen.train()
    z, vleaves, glob = make_z(g, dev, requires_grad=True)
    aux = {}
    pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                        clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                        _rng=_dropout_rng(g, dev), _aux=aux)
    for k in ("coarse_rgb_sigma", "fine_rgb_sigma"):
        assert rgb_sigma_err(aux[k], g[k]) < TOL, k
    assert scaled_err(pixels.detach().cpu().numpy(), g["pixels"]) < TOL
    assert scaled_err(depth.detach().cpu().numpy(), g["depth"]) < TOL
    loss = pixels.square().mean() + depth.mean()
    loss.backward()
    floor = reference_grad_noise_floor(g)
    tol = lambda k: max(2e-3, 2.5 * floor[k])
    assert scaled_err(vleaves[0].grad.cpu().numpy(), g["grad_feature_volume"]) < tol("feature_volume")
    if glob is not None:
        assert scaled_err(glob.grad.cpu().numpy(), g["grad_global_feature"]) < tol("global_feature")
    for k, p in gen.named_parameters():
        assert p.grad is not None, k
        assert scaled_err(p.grad.cpu().numpy(), g["grad/" + k]) < tol(k), k
    # eval mode drops nothing: a different image
    gen.eval()
    with torch.no_grad():
        px_eval, _ = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                         clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"],
                         _rng=_dropout_rng(g, dev))
    assert (px_eval - pixels.detach()).abs().max().item() > 1e-3


@pytest.mark.parametrize("name", DROP_GOLDEN)
def test_dropout_in_kernel_decisions_equal_their_numpy_twin(golden, dev, name):
    """Without injected bytes the kernels draw the keep decisions themselves (Philox4x32-10 under cnerf_cfg.philox_seed / _offset,
    stream 4 coarse / 5 fine, one block per 4 channels).  oracle/philox.py::dropout_keep restates that layout: injecting its bytes
    must reproduce the in-kernel run bit for bit, forward and backward; the keep rate is 1 - p."""
    from oracle import philox as P
    g = golden(name)
    m = g.meta
    gen = make_generator(g, dev)
    gen.train()
    B, npi, H, p = m["B"], m["R"] * m["R"] * m["S"], m["H"], m["drop_out"]
    n_drop = sum(1 for k in gen.siren.spec.layers if k != "res")
    seed, offset = 0x1234567890ABCDEF, 77
    keep = {k: P.dropout_keep(seed, offset, sid, B * npi, n_drop, H, p) for k, sid in (("drop_coarse", 4), ("drop_fine", 5))}
    assert abs(keep["drop_coarse"].mean() - (1 - p)) < 5e-3
    base = {k: v for k, v in _dropout_rng(g, dev).items() if not k.startswith("drop_")}
    runs = []
    for rng in (dict(base, drop=(p, (seed, offset))),
                dict(base, drop=(p, (seed + 1, offset)), **{k: G(v.reshape(n_drop, B, npi, H), dev) for k, v in keep.items()})):
        z, vleaves, glob = make_z(g, dev, requires_grad=True)
        gen.zero_grad()
        px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"], clamp_mode=m["clamp"],
                     nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng)
        (px.square().mean() + dp.mean()).backward()
        runs.append((px.detach().clone(), dp.detach().clone(), {k: q.grad.clone() for k, q in gen.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for k in runs[0][2]:      # (weight gradients are reduced with float atomics: order-dependent in the last bits)
        assert scaled_err(runs[0][2][k].cpu().numpy(), runs[1][2][k].cpu().numpy()) < 1e-5, k
    # another counter offset: other decisions
    z, _, _ = make_z(g, dev)
    with torch.no_grad():
        px2, _ = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"], clamp_mode=m["clamp"],
                     nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=dict(base, drop=(p, (seed, offset + 1))))
    assert not torch.equal(px2, runs[0][0])


def test_dropout_decisions_do_not_depend_on_the_chunking(dev, golden):
    """The backward re-runs the forward per chunk of images (ops.ACT_BUDGET_BYTES); keep decisions are indexed by the point's position
    in the WHOLE call (image0 offsets of the Philox counter and of the injected bytes), so one image per chunk must give the
    gradients of all images in one chunk -- in-kernel draws and injected bytes alike."""
    import cnerf_amd
    from cnerf_amd import ops
    g = golden("short_fg_drop_small")
    m = g.meta
    assert m["B"] == 2
    gen = make_generator(g, dev)
    gen.train()
    base = {k: v for k, v in _dropout_rng(g, dev).items() if not k.startswith("drop_")}
    for rng in (dict(base, drop=(m["drop_out"], (99, 3))), _dropout_rng(g, dev)):
        res = []
        budget = ops.ACT_BUDGET_BYTES
        try:
            for b in (budget, 1):                      # 1 byte: one image per chunk
                ops.ACT_BUDGET_BYTES = b
                z, vleaves, glob = make_z(g, dev, requires_grad=True)
                gen.zero_grad()
                px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True, clamp_mode=m["clamp"],
                             nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=dict(rng))
                (px.square().mean() + dp.mean()).backward()
                res.append({"feature_volume": vleaves[0].grad.clone(), "global": glob.grad.clone(),
                            **{k: q.grad.clone() for k, q in gen.named_parameters()}})
        finally:
            ops.ACT_BUDGET_BYTES = budget
        for k in res[0]:
            assert scaled_err(res[1][k].cpu().numpy(), res[0][k].cpu().numpy()) < 1e-5, k



@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_forward_replays_from_a_hip_graph(dev, precision):
    """Every launch of a forward (folded-FiLM preparation, the two field passes, resampling, merge; in-kernel Philox draws: no RNG
    kernels) goes to the current stream through the C ABI, so torch's stream capture records it as it is: the hipGraph replay
    reproduces the eager image bit for bit, and after the inputs are overwritten in place it renders the new inputs (small renders
    are launch-bound: scripts/graph_replay.py times eager against replay)."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(21)
    B, R, S, V = 2, 16, 12, 16
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 64, 32, 4, 128).to(dev)
    gen.set_device(dev)
    gen.eval()
    gen.siren.precision = precision
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev) * 0.5, torch.randn(B, 64, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    rng = {"philox": (1234, 5)}

    def call():
        with torch.no_grad():
            return gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.5, white_back=True, _rng=rng)

    eager = [t.clone() for t in call()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()                                        # warm-up on the capture stream (packing cache, allocator)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = call()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1])
    fvol.mul_(-0.7)                                   # new inputs in the same buffers
    glob.add_(0.3)
    graph.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in out]
    fresh = call()
    assert torch.equal(replayed[0], fresh[0]) and torch.equal(replayed[1], fresh[1])
    assert not torch.equal(replayed[0], eager[0])


@pytest.mark.parametrize("shape", [dict(B=2, R=11, S=10, V=40), dict(B=3, R=37, S=23, V=12), dict(B=1, R=64, S=24, V=64)])
@pytest.mark.parametrize("variant", ["SHORTSIREN_FG", "SHORTSIREN_FG_Pyrmd"])
def test_sorted_patch_scatter_matches_the_chain_scatter(dev, monkeypatch, shape, variant):
    """The feature-volume gradient of the half-precision backward is added to the volume by scatter_sorted_kernel (scatter_patch.hip:
    8 x 8-pixel patches x depth bins, corner records sorted by voxel in LDS, one add per run) from the input-tile gradients the chain
    stores.  CNERF_SCATTER=chain makes the chain add its tiles itself (the path explicit points take): the same addends in another
    order, so the volumes agree to fp32 summation noise.  Shapes: pixels ~3 voxels apart in a 40-voxel volume (most patches outgrow the
    8-voxel window: the direct path), ragged patches / a ragged last quad / several images in a 12-voxel volume (everything inside one
    window, long runs), and a 64-voxel volume at 64 x 64 rays; oblique cameras; a single level and the three-level pyramid."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    B, R, S, V = (shape[k] for k in "BRSV")
    torch.manual_seed(R)
    np.random.seed(R)
    lv = [(32, V), (64, max(V // 2, 2)), (32, max(V // 4, 2))] if variant.endswith("Pyrmd") else [(32, V)]      # the pyramid's (channels, edge) per level
    gen = ImplicitGenerator3d(variant, 32, sum(c for c, _ in lv), 4, 64).to(dev)
    gen.set_device(dev)
    gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
    gen.train()
    vols = [torch.randn(B, c, v, v, v, device=dev, requires_grad=True) for c, v in lv]
    glob = torch.randn(B, 32, device=dev)
    cam = create_cam2world_matrix(sample_camera_positions(dev, "y", 0.9, 1.1, n=B), "y", device=dev)

    def grads():
        for v in vols:
            v.grad = None
        torch.manual_seed(11)
        z = (vols[0] if len(vols) == 1 else list(vols), glob)
        px, dp = gen(z, cam, R, 49.134342641202636, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=1.0, white_back=True)
        (px.square().mean() + dp.mean()).backward()
        return [v.grad.detach().double().cpu() for v in vols]

    monkeypatch.setenv("CNERF_SCATTER", "chain")
    ref = grads()
    monkeypatch.setenv("CNERF_SCATTER", "coarse")       # the coarse pass through the sorted scatter, the fine pass by the chain
    coarse = grads()
    monkeypatch.delenv("CNERF_SCATTER")                  # the default: both ray passes sorted (the fine pass: depth bins, queued candidates)
    default = grads()
    for got in (coarse, default):
        for g, r_ in zip(got, ref):
            assert r_.abs().max() > 0
            # fp32 sums of the same addends in two orders: 2e-7 .. 4e-7 apart in the 40- and 64-voxel volumes, ~2e-6 in the pyramid's 3-voxel level
            # (each of its 27 voxels sums ~2e4 addends); a single point lost or added twice would show at 1e-3 or more
            assert ((g - r_).norm() / r_.norm()).item() < 2e-5
            assert ((g - r_).abs().max() / r_.abs().max()).item() < 1e-4


def test_half_precision_backward_reports_clamped_outliers(dev):
    """The fp16 backward stores d loss / d (sine argument) scaled by a power of two per matrix that comes from a SAMPLED maximum (every
    k-th tile group once there are >= 4096 of them: here 8192 groups, every 4th) with a factor 32 of headroom; anything beyond is
    clamped to fp16's range.  That clamp must not be silent: cnerf_render_backward counts the (tile, matrix) blocks it
    clamped in, ops.LAST_SATURATED / trainer.last["render_bwd_clamped_blocks"] surface the count.  A ray whose gradient is 1e12 x the
    others', planted in a tile group the sample does not visit (group 1 of XCD class 0 = tiles 4..7 = rays 2, 3), is reported; the same
    render without it reports nothing."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    torch.manual_seed(3)
    B, R, S, V, H = 1, 128, 64, 16, 64
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, H).to(dev)
    gen.set_device(dev)
    gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
    gen.train()
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 20
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev), torch.randn(B, 32, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).contiguous()
    cam[:, 2, 3] = -1.0
    counts = []
    for plant in (0.0, 1e7):
        fv = fvol.clone().requires_grad_(True)
        gen.zero_grad()
        px, dp = gen((fv, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus", nerf_noise=0.0, white_back=True)
        (px.square().mean() + dp.mean() + plant * px[0, :, 0, 2].sum()).backward()
        counts.append(int(ops.LAST_SATURATED.item()))
        assert torch.isfinite(fv.grad).all()
    assert counts[0] == 0 and counts[1] > 0, counts


@pytest.mark.parametrize("precision", ["fp16x3", "fp16", "fp32"])
def test_forward_is_deterministic_over_many_launches(dev, precision):
    """Forty forwards of the same inputs and draws, a fresh network (fresh weight buffers: cold in L2 at its first launch) every ten, must
    agree bit for bit.  Regression test of round 3's find: the fp16 kernels publish their weight units -- LDS-DMA copies -- to the block
    behind a barrier, and the copy's completion was left to the release fence of __syncthreads(), which the compiler does not turn into
    a wait for LDS-DMA at every call site; a block whose copy was slow (typically the first tile group of a launch) multiplied by a
    half-copied unit: whole 32-point tiles wrong in ~1 of 300 launches at 256x256x96.  The waits are written out now (wait_vmcnt)."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    R, S, V, B = 128, 64, 32, 2
    torch.manual_seed(1)
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev), torch.randn(B, 256, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(B, R * R, S, device=dev), "u_fine": torch.rand(B, R * R, S, device=dev)}
    ref = None
    for net in range(4):
        torch.manual_seed(2)
        gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
        gen.set_device(dev)
        gen.siren.precision = precision
        with torch.no_grad():
            gen.siren.final_layer.weight[3] *= 40
        for it in range(10):
            with torch.no_grad():
                px, dp = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng)
            if ref is None:
                ref = (px.clone(), dp.clone())
            else:
                assert torch.equal(px, ref[0]) and torch.equal(dp, ref[1]), (net, it, int((px != ref[0]).sum()))


@pytest.mark.parametrize("precision", ["fp16x3", "fp16"])
def test_per_point_film_is_deterministic_over_many_launches(dev, precision):
    """The same regression for the per-point FiLM family's fp16 kernels, whose weight units run through a three-slot ring behind COUNTED
    vmcnt waits (csrc/field_pw16.hip, chain_pw16.hip): twenty forwards of a TALLSIREN (fresh weight buffers every five) agree bit for
    bit, and so do the forwards of four training steps (the activation-storing instantiation, stores in flight across the barriers);
    the gradients of those steps agree to the order of fp32 atomics (volume scatter, weight reductions: not bit-reproducible by design)."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    R, S, V, B = 64, 33, 16, 2                         # 33 samples: ragged last tiles, idle waves in the last tile group
    torch.manual_seed(1)
    fvol = torch.randn(B, 32, V, V, V, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(B, R * R, S, device=dev), "u_fine": torch.rand(B, R * R, S, device=dev)}
    ref = None
    for net in range(4):
        torch.manual_seed(2)
        gen = ImplicitGenerator3d("TALLSIREN", 32, 3, 4, 256).to(dev)
        gen.set_device(dev)
        gen.siren.precision = precision
        with torch.no_grad():
            gen.siren.final_layer.weight[3] *= 40
        for it in range(5):
            with torch.no_grad():
                px, dp = gen(fvol, cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng)
            if ref is None:
                ref = (px.clone(), dp.clone())
            else:
                assert torch.equal(px, ref[0]) and torch.equal(dp, ref[1]), (net, it, int((px != ref[0]).sum()))
    if precision != "fp16x3":
        return
    gen.siren.backward_precision = "fp16"
    gen.train()
    gref = None
    for it in range(4):
        fv = fvol.clone().requires_grad_(True)
        for p_ in gen.parameters():
            p_.grad = None
        px, dp = gen(fv, cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng)
        assert torch.equal(px, ref[0]) and torch.equal(dp, ref[1]), ("storing forward", it, int((px != ref[0]).sum()))
        (px.square().mean() + dp.mean()).backward()
        grads = [fv.grad] + [p_.grad for p_ in gen.parameters()]
        if gref is None:
            gref = [g_.clone() for g_ in grads]
        else:
            for a_, b_ in zip(grads, gref):
                assert (a_ - b_).norm() <= 1e-5 * b_.norm() + 1e-30, (it, float((a_ - b_).norm() / b_.norm()))


def test_per_point_film_backward_does_not_depend_on_the_chunking(dev, monkeypatch):
    """The half-precision backward of the per-point FiLM family over three images: kept activations (one chunk), re-computed activations in
    one chunk, and re-computed image by image (cnerf_render_backward's chunk loop: per-chunk image offsets of cameras, draws, saved
    rgb_sigma, activation / gradient buffers) give the same gradients -- parameter gradients to the order of fp32 atomics, and to the order of
    the per-slab fp16 scales (sampled per chunk), i.e. 1e-3 relative L2."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    B, R, S, V = 3, 9, 13, 8
    torch.manual_seed(5)
    gen = ImplicitGenerator3d("TALLSIREN", 32, 3, 4, 128).to(dev)
    gen.set_device(dev)
    gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
    gen.train()
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 20
    fvol = torch.randn(B, 32, V, V, V, device=dev) * 0.5
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    cam[1, 0, 3], cam[2, 1, 3] = 0.1, -0.1                       # different cameras per image: a wrong per-chunk offset shows
    rng = {"u_strat": torch.rand(B, R * R, S, device=dev), "u_fine": torch.rand(B, R * R, S, device=dev)}

    def grads():
        fv = fvol.clone().requires_grad_(True)
        for p_ in gen.parameters():
            p_.grad = None
        px, dp = gen(fv, cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="softplus", nerf_noise=0.0, white_back=True, _rng=rng)
        ((px * torch.arange(1, B + 1, device=dev).view(B, 1, 1, 1)).square().mean() + dp.mean()).backward()
        return [fv.grad.clone()] + [p_.grad.clone() for p_ in gen.parameters()]

    kept = grads()
    monkeypatch.setattr(ops, "resident_act16", lambda *a, **k: None)
    one_chunk = grads()
    real_chunk = ops.backward_chunk
    seen = []

    def by_image(cfg, code, nb_max, have_act16, d):
        need = C.c_size_t(0)
        L.check(L.lib().cnerf_backward_workspace_bytes(C.byref(cfg), code, 1, 0, C.byref(need)), "cnerf_backward_workspace_bytes")
        seen.append(nb_max)
        return 1, need.value
    import ctypes as C
    L = cnerf_amd._lib
    monkeypatch.setattr(ops, "backward_chunk", by_image)
    chunked = grads()
    assert seen == [B]
    monkeypatch.setattr(ops, "backward_chunk", real_chunk)
    for a_, b_, c_ in zip(kept, one_chunk, chunked):
        n = a_.norm().item()
        assert n > 0
        assert (a_ - b_).norm().item() <= 1e-5 * n, ((a_ - b_).norm().item() / n)
        assert (a_ - c_).norm().item() <= 1e-3 * n, ((a_ - c_).norm().item() / n)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_config4_per_gpu_batch_is_batch_invariant(dev, precision):
    """BASELINE config 4's per-GPU workload -- 8 images of 128x128 rays x (64 + 64) samples, 64^3 volumes, hidden 256, what bench.py
    times -- checked by a size-independent property (the oracle takes 7 s per image): every image of the batch-8 render equals, bit
    for bit, the same image rendered alone with its own slice of the draws.  Images are independent units of the path (one
    feature volume, one FiLM vector, one camera each); what could couple them is exactly what a batch changes -- tile groups spanning an
    image boundary, the per-image FiLM constants / weight copies restaged in LDS at the image switch, the XCD banding of eight images."""
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    torch.manual_seed(4)
    np.random.seed(4)
    B, R, S, V = 8, 128, 64, 64
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
    gen.set_device(dev)
    gen.siren.precision = precision
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 30
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev) * 0.5, torch.randn(B, 256, device=dev)
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y").to(dev)
    P = R * R
    rng = {"u_strat": torch.rand(B, P, S, device=dev), "eps_coarse": torch.randn(B, P, S, device=dev), "u_fine": torch.rand(B, P, S, device=dev),
           "eps_final": torch.randn(B, P, 2 * S, device=dev)}
    meta = dict(clamp_mode="softplus", nerf_noise=0.5, white_back=True)
    with torch.no_grad():
        px, dp = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, _rng=rng, **meta)
        assert torch.isfinite(px).all() and torch.isfinite(dp).all()
        for i in (0, 3, 7):
            one = {k: v[i:i + 1].contiguous() for k, v in rng.items()}
            pi, di = gen((fvol[i:i + 1].contiguous(), glob[i:i + 1].contiguous()), cam[i:i + 1].contiguous(), R, 49.13, 0.25, 1.95, S, True, _rng=one, **meta)
            assert torch.equal(pi[0], px[i]) and torch.equal(di[0], dp[i]), i
