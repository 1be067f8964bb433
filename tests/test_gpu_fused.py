"""The fused per-point frame step (glh_step, glimpse_amd/csrc/glh_point.h) against (i) the staged
kernels it replaces -- same particles and weights BIT FOR BIT, since both apply the same
arithmetic to the same draws -- and (ii) the reference's golden posteriors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    from glimpse_amd import _lib

    assert _lib.device_count() >= 1
    return _lib


def _run_golden(g, fused):
    """Frame loop of tracker.py:326-357 with glh_step wherever no template starts mid-sequence."""
    from tests.helpers_gpu import batched_draws, context_for

    ctx = context_for(g, debug=False)
    ctx.set_fused(fused)
    matching = g["matching"]
    taus = np.diff(g["datetimes_days"])
    T, O = matching.shape
    template_indices = (matching >= 0).argmax(axis=0)
    init, ev, us = batched_draws(g)
    ctx.set_frame(0)
    ctx.init_particles(normals=init)
    for o in np.nonzero(template_indices == 0)[0]:
        ctx.init_templates(int(o), int(matching[0][o]))
    ctx.record_moments(0)
    for i in range(1, T):
        late = np.nonzero(template_indices == i)[0]
        if len(late):
            ctx.set_frame(i)
            ctx.evolve(taus[i - 1], normals=ev[i - 1])
            for o in late:
                ctx.init_templates(int(o), int(matching[i][o]))
            ctx.update_weights(matching[i])
            ctx.resample(u=us[i - 1])
            ctx.record_moments(i)
        else:
            ctx.step(i, taus[i - 1], matching[i], normals=ev[i - 1], u=us[i - 1])
    out = dict(moments=ctx.get_moments(0, T), particles=ctx.get_particles(), weights=ctx.get_weights(),
               status=ctx.point_status(), obs_status=ctx.observer_status())
    ctx.close()
    return out


@pytest.mark.parametrize("name", ["g8_c1.npz", "g8_c2mini.npz", "g8_c5mini.npz"])
def test_fused_step_equals_staged_and_reference(lib, golden, name):
    g = golden(name)
    fused = _run_golden(g, 1)
    staged = _run_golden(g, 0)
    hbm = _run_golden(g, 2)  # same kernel with every tile in the HBM workspaces (large-tile path)
    ok = ~g["errors"].astype(bool)
    np.testing.assert_array_equal(hbm["particles"][ok], staged["particles"][ok])
    np.testing.assert_array_equal(hbm["weights"][ok], staged["weights"][ok])
    np.testing.assert_array_equal(hbm["moments"][:, ok], fused["moments"][:, ok])
    np.testing.assert_array_equal(fused["status"], staged["status"])
    np.testing.assert_array_equal(fused["obs_status"], staged["obs_status"])
    np.testing.assert_array_equal(fused["particles"][ok], staged["particles"][ok])
    np.testing.assert_array_equal(fused["weights"][ok], staged["weights"][ok])
    np.testing.assert_allclose(fused["moments"][:, ok], staged["moments"][:, ok], rtol=1e-12, atol=1e-13)
    means = np.transpose(fused["moments"][:, :, 0:6], (1, 0, 2))
    sigmas = np.transpose(fused["moments"][:, :, 6:12], (1, 0, 2))
    np.testing.assert_allclose(means[ok], g["means"][ok], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(sigmas[ok], g["out_sigmas"][ok], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("cfg", [("C2", 16, 2000, 1), ("C3", 6, 5000, 1), ("C5", 4, 3001, 1), ("C3", 3, 10240, 1),
                                 ("C3", 2, 12000, 1), ("C2", 5, 777, 3), ("C5", 3, 1500, 3)])
def test_fused_step_equals_staged_with_device_rng(lib, cfg):
    """Philox mode (what large runs use): re-evolving the gathered sources from the counter-based
    noise gives exactly the state the staged kernels store and gather."""
    from glimpse_amd import workloads

    name, P, N, channels = cfg
    T = 5
    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [[wl.scene.render(wl.cams[o], float(t), channels=channels) for t in range(T)] for o in range(wl.O)]
    res = []
    for mode in (1, 0, 2):
        with lib.Context(wl.P, wl.N, wl.O, max_search_dim=160, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames, channels=channels)
            ctx.set_fused(mode)
            ctx.set_frame(0)
            ctx.init_particles(seed=11)
            for o in range(wl.O):
                ctx.init_templates(o, 0)
            ctx.record_moments(0)
            for i in range(1, T):
                ctx.step(i, 1.0, [i] * wl.O, seed=11)
            assert (ctx.observer_status() == lib.OBS_OK).all()
            assert (ctx.point_status() == 0).all()
            res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T)))
    for other in (1, 2):
        np.testing.assert_array_equal(res[0][0], res[other][0])
        np.testing.assert_array_equal(res[0][1], res[other][1])
        np.testing.assert_allclose(res[0][2], res[other][2], rtol=1e-12, atol=1e-13)
    # the filter follows the synthetic motion (0.15 units/frame along x)
    vx = res[0][2][-1, :, 3]
    assert abs(np.median(vx) - 0.15) < 0.05


@pytest.mark.parametrize("cfg", [("C2", 16, 2000, 1, 1), ("C3", 6, 5000, 1, 1), ("C5", 4, 3001, 1, 1),
                                 ("C3", 3, 10240, 1, 1), ("C3", 2, 12000, 1, 1), ("C5", 3, 1500, 3, 1),
                                 ("C3", 5, 5000, 1, 2), ("C3", 700, 512, 1, 1)])
def test_track_equals_the_same_steps(lib, cfg):
    """glh_track (the frame loop of tracker.py:326-357 in one call) leaves exactly what the same glh_step calls
    leave: particles, weights, every row of the moments history, statuses.  Includes
    frames an observer skips (image None), tiles forced to the HBM workspaces, non-unit and negative time steps,
    and more tracks than resident workgroups."""
    from glimpse_amd import workloads

    name, P, N, channels, mode = cfg
    T = 7
    wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [[wl.scene.render(wl.cams[o], float(t), channels=channels) for t in range(T)] for o in range(wl.O)]
    images = [[i] * wl.O for i in range(1, T)]
    if wl.O > 1:
        images[2][1] = None  # observer 1 has no image matching frame 3
    taus = [1.0] * (T - 1)
    res = []
    for how in ("steps", "track", "split"):
        with lib.Context(wl.P, wl.N, wl.O, max_search_dim=160, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames, channels=channels)
            ctx.set_fused(mode)
            ctx.set_frame(0)
            ctx.init_particles(seed=11)
            for o in range(wl.O):
                ctx.init_templates(o, 0)
            ctx.record_moments(0)
            fr = list(range(1, T))
            if how == "steps":
                for i in fr:
                    ctx.step(i, taus[i - 1], images[i - 1], seed=11)
            elif how == "track":
                ctx.track(fr, taus, images, seed=11)
            else:  # an odd and an even number of frames per call, then a single step
                ctx.track(fr[:3], taus[:3], images[:3], seed=11)
                ctx.track(fr[3:5], taus[3:5], images[3:5], seed=11)
                ctx.step(fr[5], taus[5], images[5], seed=11)
            res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), ctx.point_status(),
                        ctx.observer_status(), ctx.search_boxes()))
    assert (res[0][3] == 0).all()
    for other in (1, 2):
        for k in range(6):
            np.testing.assert_array_equal(res[0][k], res[other][k])


def test_sharded_contexts_reproduce_the_unsharded_run(lib):
    """Device RNG is keyed on the GLOBAL point index (glh_set_point_offset): two contexts tracking
    points [0, 5) and [5, 8) give exactly the particles of one context tracking all 8."""
    from glimpse_amd import sharding, workloads

    T, P, N = 4, 8, 1500
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(0)]

    def run(lo, hi):
        with lib.Context(hi - lo, N, 1, max_search_dim=160, max_frames=T) as ctx:
            ctx.observer_init(0, T, 640, 640, 1, wl.sigmas[0])
            ctx.observer_set_cameras(0, np.tile(wl.cams[0], (T, 1)))
            for t in range(T):
                ctx.observer_upload_frame(0, t, frames[0][t])
            ctx.begin_sequence(hi - lo, N, wl.tile)
            ctx.set_motion_cartesian(wl.params[lo:hi])
            ctx.set_point_offset(lo)
            ctx.set_frame(0)
            ctx.init_particles(seed=5)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            for i in range(1, T):
                ctx.step(i, 1.0, [i], seed=5)
            return ctx.get_particles(), ctx.get_moments(0, T)

    full_p, full_m = run(0, P)
    parts = [run(*sharding.shard_range(P, 2, r)) for r in range(2)]
    np.testing.assert_array_equal(np.concatenate([p for p, _ in parts]), full_p)
    np.testing.assert_array_equal(np.concatenate([m for _, m in parts], axis=1), full_m)


def test_fused_step_edge_cases_match_staged(lib):
    """Failure modes inside the fused kernel behave like the staged kernels: a NaN particle
    (ValueError bit, tracker.py:118 -> uv NaN -> search box out of bounds, tracker.py:597-601), a cloud that
    leaves the image (warning + skip: weights come from the motion model alone) and a template box
    outside the image (IndexError bit, raster.py:417; no template -> observer skipped)."""
    from glimpse_amd import workloads

    T, P, N = 4, 6, 1200
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(0)]
    params = wl.params.copy()
    params[1, 0:2] = (-30.6, 0.0)   # ~14 px from the left edge: template fits, the evolved cloud does not
    params[2, 0:2] = (-31.6, 5.0)   # template box itself leaves the image
    params[5, 17] = 0.5             # dem_sigma > 0: CartesianMotion.compute_log_likelihoods contributes
    out = []
    for mode in (1, 0, 2):
        with lib.Context(P, N, 1, max_search_dim=160, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames)
            ctx.set_motion_cartesian(params)
            ctx.set_fused(mode)
            ctx.set_frame(0)
            ctx.init_particles(seed=3)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            p0 = ctx.get_particles()
            p0[4, 17, 0] = np.nan  # one missing value in point 4
            p0[5, 3, 2] = np.nan   # and one in point 5, whose DEM term turns it into a NaN weight
            ctx.set_particles(p0)
            for i in range(1, T):
                ctx.step(i, 1.0, [i], seed=3)
            out.append(dict(p=ctx.get_particles(), w=ctx.get_weights(), st=ctx.point_status(),
                            ob=ctx.observer_status(), ef=ctx.point_error_frame(), m=ctx.get_moments(0, T)))
    ref = out[1]
    assert ref["st"][4] & lib.PT_NAN and ref["st"][2] & lib.PT_TEMPLATE_OOB
    assert ref["ob"][0, 2] == lib.OBS_NO_TEMPLATE
    assert ref["ob"][0, 1] == lib.OBS_OUT_OF_BOUNDS or ref["st"][1] == 0
    for o in (out[0], out[2]):
        np.testing.assert_array_equal(o["st"], ref["st"])
        np.testing.assert_array_equal(o["ob"], ref["ob"])
        np.testing.assert_array_equal(o["ef"], ref["ef"])
        np.testing.assert_array_equal(o["p"], ref["p"])
        np.testing.assert_array_equal(o["w"], ref["w"])
        ok = [0, 1, 2, 3]
        np.testing.assert_allclose(o["m"][:, ok], ref["m"][:, ok], rtol=1e-12, atol=1e-13)
    # NaN weights (point 5): np.searchsorted returns n for every position -> IndexError in the reference; here the
    # positions are clamped to the last source and the point is flagged
    assert ref["st"][5] & lib.PT_NAN and ref["st"][5] & lib.PT_RESAMPLE_CLAMP


def test_fused_step_with_gridded_surfaces_equals_staged(lib):
    """Gridded dem / dem_sigma (bilinear Raster.sample per particle, raster.py:913-1027) and a viewshed inside
    the fused kernel (its SURF variant) against the staged kernels, device RNG."""
    import glimpse_amd
    from glimpse_amd import workloads

    T, P, N = 4, 5, 1300
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(0)]
    rng = np.random.default_rng(9)
    lim = (-40.0, 40.0)
    dem = glimpse_amd.Raster(0.05 * rng.standard_normal((33, 41)), x=lim, y=(40.0, -40.0))
    dem_sigma = glimpse_amd.Raster(0.3 + 0.1 * rng.random((17, 19)), x=lim, y=lim)
    vis = np.ones((16, 16))
    vis[:, :2] = 0  # western strip hidden: points there raise the NOT_VISIBLE bit
    viewshed = glimpse_amd.Raster(vis, x=lim, y=lim)
    params = np.zeros((P, lib.MOTION_FULL_LEN))
    params[:, :18] = wl.params
    params[:, 7:10] = (0.2, 0.2, 0.05)
    params[:, 13:16] = (0.05, 0.05, 0.01)
    params[:, 20] = 1.0
    params[:, 21] = 1.0
    params[3, 0:2] = (-31.0, 3.0)  # inside the hidden strip (and far from the image border)
    out = []
    for mode in (1, 0):
        with lib.Context(P, N, 1, max_search_dim=160, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames)
            ctx.set_raster(lib.RASTER_DEM, dem)
            ctx.set_raster(lib.RASTER_DEM_SIGMA, dem_sigma)
            ctx.set_raster(lib.RASTER_VIEWSHED, viewshed)
            ctx.set_motion(params)
            ctx.set_fused(mode)
            ctx.set_frame(0)
            ctx.init_particles(seed=21)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            for i in range(1, T):
                ctx.step(i, 1.0, [i], seed=21)
            out.append(dict(p=ctx.get_particles(), w=ctx.get_weights(), st=ctx.point_status(),
                            ef=ctx.point_error_frame(), m=ctx.get_moments(0, T)))
    fused, staged = out
    assert staged["st"][3] & lib.PT_NOT_VISIBLE and staged["st"][0] == 0
    np.testing.assert_array_equal(fused["st"], staged["st"])
    np.testing.assert_array_equal(fused["ef"], staged["ef"])
    np.testing.assert_array_equal(fused["p"], staged["p"])
    np.testing.assert_array_equal(fused["w"], staged["w"])
    np.testing.assert_allclose(fused["m"], staged["m"], rtol=1e-12, atol=1e-13)
    # the DEM term acts: z stays near the surface (sigma ~0.3) although vz noise accumulates
    assert np.abs(fused["m"][-1, 0, 2]) < 1.0


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("gridded", [False, True])
def test_fused_step_evolves_every_motion_model_like_the_staged_kernels(gridded, math):
    """CylindricalMotion and the tangent models (motion.py:207-522) in the fused kernel's general instantiation: one
    context mixing all four kinds (the kind is a per-point parameter), constant and gridded surfaces, a frame on
    which no observer has an image (the tangent models have no log-likelihood term either: the weights stay as they
    are, tracker.py:146-149) -- bit for bit the staged kernels, host-fed draws and device draws."""
    import glimpse_amd
    from glimpse_amd import _lib as lib
    from glimpse_amd import workloads

    T, P, N = 5, 8, 1500
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
    frames = [wl.frames(0)]
    rng = np.random.default_rng(4)
    lim = (-40.0, 40.0)
    dem = glimpse_amd.Raster(0.05 * rng.standard_normal((33, 41)), x=lim, y=(40.0, -40.0))
    dem_sigma = glimpse_amd.Raster(0.3 + 0.1 * rng.random((17, 19)), x=lim, y=lim)
    params = np.zeros((P, lib.MOTION_FULL_LEN))
    params[:, :18] = wl.params
    params[:, 18] = np.arange(P) % 4  # cartesian, cylindrical, tangent cartesian, tangent cylindrical, ...
    params[:, 19] = 0.05              # slope_sigma of the tangent models
    params[:, 17] = 0.4               # constant dem_sigma (the Cartesian / Cylindrical likelihood term)
    for p in range(P):
        if params[p, 18] in (1, 3):   # (vr, theta, vz) instead of (vx, vy, vz)
            params[p, 4:7] = (0.15, 0.0, 0.0)
            params[p, 7:10] = (0.05, 0.3, 0.02)
            params[p, 13:16] = (0.02, 0.05, 0.005)
        else:
            params[p, 7:10] = (0.2, 0.2, 0.02)
            params[p, 13:16] = (0.05, 0.05, 0.005)
    if gridded:
        params[:, 20] = 1.0
        params[:, 21] = 1.0
    images = [[1], [2], [-1], [4]]    # frame 3: no image
    host = np.random.default_rng(12)
    init, ev, us = host.standard_normal((P, N, 6)), host.standard_normal((T - 1, P, N, 3)), host.random((T - 1, P))
    for device_rng in (False, True):
        out = []
        for mode in (1, 0):
            with lib.Context(P, N, 1, max_search_dim=160, max_frames=T) as ctx:
                workloads.setup_context(ctx, wl, frames)
                if gridded:
                    ctx.set_raster(lib.RASTER_DEM, dem)
                    ctx.set_raster(lib.RASTER_DEM_SIGMA, dem_sigma)
                ctx.set_motion(params)
                ctx.set_fused(mode)
                ctx.set_math(math)  # (fast: the general instantiation in fast arithmetic, glh_set_math)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=8) if device_rng else ctx.init_particles(normals=init)
                ctx.init_templates(0, 0)
                ctx.record_moments(0)
                ctx.profile_enable(True)
                idx = []
                for i in range(1, T):
                    if device_rng:
                        ctx.step(i, 1.0, images[i - 1], seed=8)
                    else:
                        ctx.step(i, 1.0, images[i - 1], normals=ev[i - 1], u=us[i - 1])
                    idx.append(ctx.resample_indices())
                stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
                assert ("point_step" in stages) == (mode == 1) and ("resample" in stages) == (mode == 0)
                out.append(dict(p=ctx.get_particles(), w=ctx.get_weights(), st=ctx.point_status(), idx=np.stack(idx),
                                m=ctx.get_moments(0, T), obs=ctx.observer_status_frames(1, T - 1)))
        fused, staged = out
        assert (staged["st"] == 0).all()
        np.testing.assert_array_equal(fused["st"], staged["st"])
        np.testing.assert_array_equal(fused["obs"], staged["obs"])
        assert (fused["obs"][2] == lib.OBS_SKIPPED).all() and (fused["obs"][3] == lib.OBS_OK).all()
        np.testing.assert_array_equal(fused["idx"], staged["idx"])
        np.testing.assert_array_equal(fused["p"], staged["p"])
        np.testing.assert_array_equal(fused["w"], staged["w"])
        np.testing.assert_allclose(fused["m"], staged["m"], rtol=1e-11, atol=1e-12)
        # tangent models leave vz = 0 and follow the surface
        tangent = params[:, 18] >= 2
        assert (fused["p"][tangent][..., 5] == 0).all()


def _multi_observer_case(O, T=4, P=3, N=1500, seed=5):
    """O stations around one scene (a nadir camera with k1-k3, the oblique station of C5, a second nadir station off to
    the side with another focal length, a second oblique station), points every one of them sees."""
    from glimpse_amd import synth

    imgsz = (512, 512)
    cams = [synth.nadir_camera(imgsz, f=1000.0, height=100.0, k=(0.05, -0.01, 0.002, 0, 0, 0)),
            synth.pack_camera(imgsz=imgsz, f=1200.0, k=(0.03, 0, 0), xyz=(40, -30, 90), viewdir=(-53.13, -60.9, 0)),
            synth.nadir_camera(imgsz, f=850.0, height=110.0, k=(0.02, 0, 0, 0, 0, 0), xyz_offset=(4.0, -3.0)),
            synth.pack_camera(imgsz=imgsz, f=1100.0, k=(0, 0, 0), xyz=(-35, 25, 95), viewdir=(125.5, -65.6, 0))][:O]
    scene = synth.default_scene(cams[0], seed=seed, velocity=(0.15, 0.0), n_frames=T, margin=60.0)
    frames = [[scene.render(cam, float(t)) for t in range(T)] for cam in cams]
    rng = np.random.default_rng(seed)
    xy = []
    while len(xy) < P:
        cand = rng.uniform(-6, 6, 2)
        uv = [synth.project(cam, np.array([[cand[0], cand[1], 0.0]]))[0] for cam in cams]
        if all(90 < u[0] < imgsz[0] - 90 and 90 < u[1] < imgsz[1] - 90 for u in uv):
            xy.append(cand)
    params = np.zeros((P, 18))
    params[:, 0:2] = xy
    params[:, 2:4] = 0.15
    params[:, 4:7] = (0.15, 0.0, 0.0)
    params[:, 7:10] = (0.1, 0.1, 0.03)
    params[:, 13:16] = (0.04, 0.04, 0.01)
    params[:, 17] = 0.4
    return dict(cams=cams, frames=frames, params=params, imgsz=imgsz, T=T, P=P, N=N, sigmas=[0.3, 0.45, 0.35, 0.5][:O])


@pytest.mark.parametrize("O", [3, 4])
@pytest.mark.parametrize("math", ["exact", "fast"])
def test_three_and_four_observers_on_the_fused_kernel(lib, O, math):
    """Up to MAX_OBS = 4 observers per point run on the fused kernel (k_point_step<.., 0, O, ..>): the log likelihoods of
    all stations add up (tracker.py:139-146).  Bit for bit the staged kernels (also with every tile forced through the
    HBM workspaces), and the oracle's indices and posteriors on the same host-fed draws."""
    from oracle import motion as omotion
    from oracle import tracker as otracker

    cs = _multi_observer_case(O)
    P, N, T = cs["P"], cs["N"], cs["T"]
    rng = np.random.default_rng(O)
    init = rng.standard_normal((P, N, 6))
    ev = rng.standard_normal((T - 1, P, N, 3))
    us = rng.random((T - 1, P))
    res = {}
    # (mode 2 -- every tile through the HBM workspaces -- has its own LDS plan and with it its own bound on the surfaces
    # the fast arithmetic samples in per-cell form: comparable bit for bit in exact arithmetic only)
    modes = (1, 0, 2) if math == "exact" else (1, 0)
    for mode in modes:
        with lib.Context(P, N, O, max_tile=31, max_search_dim=160, max_frames=T) as ctx:
            for o in range(O):
                ctx.observer_init(o, T, cs["imgsz"][0], cs["imgsz"][1], 1, cs["sigmas"][o])
                ctx.observer_set_cameras(o, np.tile(cs["cams"][o], (T, 1)))
                for t in range(T):
                    ctx.observer_upload_frame(o, t, cs["frames"][o][t])
            ctx.begin_sequence(P, N, (21, 21))
            ctx.set_motion_cartesian(cs["params"])
            ctx.set_math(math)
            ctx.set_fused(mode)
            ctx.set_debug(2)
            ctx.set_frame(0)
            ctx.init_particles(normals=init)
            for o in range(O):
                ctx.init_templates(o, 0)
            ctx.record_moments(0)
            idx = []
            for i in range(1, T):
                # (frame 2: the last station has no image)
                images = [i] * O if i != 2 else [i] * (O - 1) + [-1]
                ctx.step(i, 1.0, images, normals=ev[i - 1], u=us[i - 1])
                idx.append(ctx.resample_indices())
                if mode:
                    assert ctx.last_variant()[:3] == (512, 0, O)
            assert (ctx.point_status() == 0).all()
            res[mode] = dict(particles=ctx.get_particles(), weights=ctx.get_weights(), moments=ctx.get_moments(0, T),
                             idx=np.stack(idx))
    for other in modes[1:]:
        np.testing.assert_array_equal(res[1]["idx"], res[other]["idx"])
        np.testing.assert_array_equal(res[1]["particles"], res[other]["particles"])
        np.testing.assert_array_equal(res[1]["weights"], res[other]["weights"])
        np.testing.assert_allclose(res[1]["moments"], res[other]["moments"], rtol=1e-12, atol=1e-13)
    # (the SSD restated with the kernels' accumulation: index-for-index comparisons are made against that one)
    observers = [otracker.Observer(cs["frames"][o], np.tile(cs["cams"][o], (T, 1)), cs["sigmas"][o], ssd="row_f32")
                 for o in range(O)]
    matching = np.tile(np.arange(T)[:, None], (1, O))
    matching[2, O - 1] = -1
    for p in range(P):
        q = cs["params"][p]
        model = omotion.CartesianMotion(xy=q[0:2], xy_sigma=q[2:4], vxyz=q[4:7], vxyz_sigma=q[7:10], axyz=q[10:13],
                                        axyz_sigma=q[13:16], dem=q[16], dem_sigma=q[17], n=N)
        draws = {"init": init[p], "evolve": [ev[s, p] for s in range(T - 1)], "u": [us[s, p] for s in range(T - 1)]}
        trace = []
        ref = otracker.track_one(model, observers, matching, np.ones(T - 1), tile_size=(21, 21), draws=draws, trace=trace)
        np.testing.assert_allclose(res[1]["moments"][:, p, 0:6], ref["means"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(res[1]["moments"][:, p, 6:12], ref["sigmas"], rtol=1e-7, atol=1e-8)
        steps = [tr["idx"] for tr in trace if "idx" in tr]
        for s in range(T - 1):
            np.testing.assert_array_equal(res[1]["idx"][s][p], steps[s])


@pytest.mark.parametrize("math", ["exact", "fast"])
def test_tangent_models_over_rasters_with_three_observers(lib, math):
    """The instantiations with the raster samples where observer 0's coordinates are parked in memory (three observers:
    k_point_step<.., 0, 3, 2, ..>): the tangent models' evolved height, parked by phase A for the gather, shares the point's
    observer-0 slot of the uv scratch with those coordinates; windows of both surfaces in LDS; a Cartesian point with the
    DEM term and a point over constants in the same batch.  Bit for bit the staged kernels, host-fed and device draws."""
    import glimpse_amd

    cs = _multi_observer_case(3, P=4)
    P, N, T = cs["P"], cs["N"], cs["T"]
    rng = np.random.default_rng(31)
    xy = cs["params"][:, 0:2]
    lo, hi = xy.min(axis=0) - 60.0, xy.max(axis=0) + 60.0
    nx, ny = 57, 49
    dem = glimpse_amd.Raster(0.03 * rng.standard_normal((ny, nx)), x=(lo[0], hi[0]), y=(hi[1], lo[1]))
    dem_sigma = glimpse_amd.Raster(0.2 + 0.1 * rng.random((ny, nx)), x=(lo[0], hi[0]), y=(hi[1], lo[1]))
    params = np.zeros((P, lib.MOTION_FULL_LEN))
    params[:, :18] = cs["params"]
    params[:, 18] = [2, 3, 0, 2]  # tangent Cartesian, tangent cylindrical, Cartesian (DEM term over rasters), tangent Cartesian
    params[:, 19] = 0.05
    params[:, 20:22] = 1.0
    params[-1, 20:22] = 0.0  # (the last point: constant surfaces, in a context that holds rasters)
    params[-1, 17] = 0.3
    for p in range(P):
        if params[p, 18] == 3:
            params[p, 4:7] = (0.15, 0.0, 0.0)
            params[p, 7:10] = (0.05, 0.3, 0.0)
            params[p, 13:16] = (0.02, 0.05, 0.0)
    init = rng.standard_normal((P, N, 6))
    ev = rng.standard_normal((T - 1, P, N, 3))
    us = rng.random((T - 1, P))
    for device_rng in (False, True):
        res = []
        for mode in (1, 0):
            with lib.Context(P, N, 3, max_tile=31, max_search_dim=160, max_frames=T) as ctx:
                for o in range(3):
                    ctx.observer_init(o, T, cs["imgsz"][0], cs["imgsz"][1], 1, cs["sigmas"][o])
                    ctx.observer_set_cameras(o, np.tile(cs["cams"][o], (T, 1)))
                    for t in range(T):
                        ctx.observer_upload_frame(o, t, cs["frames"][o][t])
                ctx.begin_sequence(P, N, (21, 21))
                ctx.set_raster(lib.RASTER_DEM, dem)
                ctx.set_raster(lib.RASTER_DEM_SIGMA, dem_sigma)
                ctx.set_motion(params)
                ctx.set_math(math)
                ctx.set_fused(mode)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=5) if device_rng else ctx.init_particles(normals=init)
                for o in range(3):
                    ctx.init_templates(o, 0)
                ctx.record_moments(0)
                idx = []
                for i in range(1, T):
                    images = [i] * 3 if i != 2 else [-1, -1, -1]  # (frame 2: no image anywhere -- the weights stay)
                    if device_rng:
                        ctx.step(i, 1.0, images, seed=5)
                    else:
                        ctx.step(i, 1.0, images, normals=ev[i - 1], u=us[i - 1])
                    idx.append(ctx.resample_indices())
                    if mode:
                        assert ctx.last_variant()[:3] == (512, 0, 3) and ctx.last_variant()[3] & 8
                assert (ctx.point_status() == 0).all()
                res.append(dict(p=ctx.get_particles(), w=ctx.get_weights(), m=ctx.get_moments(0, T), idx=np.stack(idx)))
        np.testing.assert_array_equal(res[0]["idx"], res[1]["idx"])
        np.testing.assert_array_equal(res[0]["p"], res[1]["p"])
        np.testing.assert_array_equal(res[0]["w"], res[1]["w"])
        np.testing.assert_allclose(res[0]["m"], res[1]["m"], rtol=1e-12, atol=1e-13)
        assert (res[0]["p"][[0, 1, 3], :, 5] == 0.0).all()  # (the tangent models leave vz = 0)


def test_fast_arithmetic_over_rasters_stays_within_rounding_of_exact(lib):
    """The fast arithmetic's own forms of the surface samples (glh_math.h: raster_bilinear_fast / raster_sample_window --
    reciprocal interval widths, fused multiply-adds, the sample served from the LDS window alone), of the tangent models'
    step (Newton square root) and of the DEM term (Newton reciprocal) against the exact arithmetic -- scipy's bilinear form,
    IEEE divisions, the goldens' arithmetic -- on the SAME host-fed draws: the first update's particles agree to rounding,
    the posteriors of the whole sequence to 1e-7.  (Fused == staged in each arithmetic is tested above; this pins the
    fast forms themselves, on the device's own reciprocal / square-root instructions.)  One and two observers, windows
    in the raster's interior and -- the point in a corner of a small raster -- at its edge (the general code)."""
    import glimpse_amd

    # (observers, raster nodes, margin around the points: 57 x 49 -- windows in the raster's interior; 9 x 11 -- a raster smaller
    # than a window; 14 x 13 barely covering the points -- whole windows clipped to the raster's edges, samples outside their
    # served span)
    for O, (nx, ny), pad in ((1, (57, 49), 60.0), (2, (57, 49), 60.0), (1, (9, 11), 3.0), (1, (14, 13), 3.0)):
        cs = _multi_observer_case(max(O, 2), P=4)
        P, N, T = cs["P"], cs["N"], cs["T"]
        rng = np.random.default_rng(17 + O + nx)
        xy = cs["params"][:, 0:2]
        lo, hi = xy.min(axis=0) - pad, xy.max(axis=0) + pad
        dem = glimpse_amd.Raster(0.03 * rng.standard_normal((ny, nx)), x=(lo[0], hi[0]), y=(hi[1], lo[1]))
        dem_sigma = glimpse_amd.Raster(0.2 + 0.1 * rng.random((ny, nx)), x=(lo[0], hi[0]), y=(hi[1], lo[1]))
        params = np.zeros((P, lib.MOTION_FULL_LEN))
        params[:, :18] = cs["params"]
        params[:, 18] = [2, 3, 0, 2]  # tangent Cartesian, tangent cylindrical, Cartesian (DEM term over rasters), tangent Cartesian
        params[:, 19] = 0.05
        params[:, 20:22] = 1.0
        params[1, 4:7] = (0.15, 0.0, 0.0)
        params[1, 7:10] = (0.05, 0.3, 0.0)
        params[1, 13:16] = (0.02, 0.05, 0.0)
        init = rng.standard_normal((P, N, 6))
        ev = rng.standard_normal((T - 1, P, N, 3))
        us = rng.random((T - 1, P))
        res = {}
        for math in ("exact", "fast"):
            with lib.Context(P, N, O, max_tile=31, max_search_dim=160, max_frames=T) as ctx:
                for o in range(O):
                    ctx.observer_init(o, T, cs["imgsz"][0], cs["imgsz"][1], 1, cs["sigmas"][o])
                    ctx.observer_set_cameras(o, np.tile(cs["cams"][o], (T, 1)))
                    for t in range(T):
                        ctx.observer_upload_frame(o, t, cs["frames"][o][t])
                ctx.begin_sequence(P, N, (21, 21))
                ctx.set_raster(lib.RASTER_DEM, dem)
                ctx.set_raster(lib.RASTER_DEM_SIGMA, dem_sigma)
                ctx.set_motion(params)
                ctx.set_math(math)
                ctx.set_frame(0)
                ctx.init_particles(normals=init)
                for o in range(O):
                    ctx.init_templates(o, 0)
                ctx.record_moments(0)
                first = None
                for i in range(1, T):
                    ctx.step(i, 1.0, [i] * O, normals=ev[i - 1], u=us[i - 1])
                    if i == 1:
                        first = (ctx.get_particles(), ctx.get_weights())
                    assert ctx.last_variant()[3] & 8  # (the instantiation with the raster samples)
                assert (ctx.point_status() == 0).all()
                res[math] = dict(first=first, m=ctx.get_moments(0, T))
        # the first update: the same resampling decisions (the indices may differ where two cumulative weights are within
        # rounding of a position -- not on these draws), states within rounding of each other
        pe, pf = res["exact"]["first"][0], res["fast"]["first"][0]
        same = (np.abs(pe - pf) <= 1e-9 * (1.0 + np.abs(pe))).all(axis=2)
        assert same.mean() > 0.999, same.mean()
        np.testing.assert_allclose(res["fast"]["m"], res["exact"]["m"], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("variant", [dict(highpass=(3, 3)), dict(highpass=(7, 5)), dict(highpass=(1, 3)),
                                     dict(interpolation=(1, 1)), dict(highpass=(3, 3), interpolation=(1, 1)),
                                     dict(highpass=(5, 5), hp_mode="nearest"), dict(highpass=(3, 7), hp_mode="wrap"),
                                     dict(highpass=(5, 5), hp_mode="mirror")])
def test_other_median_windows_and_bilinear_sampling_on_the_fused_kernel(lib, variant, math):
    """Tracker(highpass={"size": ...}) other than 5 x 5 and Tracker(interpolation={"kx": 1, "ky": 1}) run on the
    general instantiations of the fused kernel (rounds 1-2: staged kernels only): bit for bit the staged kernels, on
    gray and RGB frames, host-fed and device draws."""
    from glimpse_amd import workloads

    T = 5
    for name, P, N, channels in (("C2", 6, 1500, 1), ("C5", 3, 2000, 3)):
        wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
        wl.channels = channels
        frames = [wl.frames(o) for o in range(wl.O)]
        rng = np.random.default_rng(7)
        ev, us = rng.standard_normal((P, N, 3)), rng.random(P)
        res = []
        for mode in (1, 0):
            with lib.Context(wl.P, wl.N, wl.O, max_search_dim=160, max_frames=T) as ctx:
                workloads.setup_context(ctx, wl, frames)
                if "highpass" in variant:
                    ctx.set_highpass(variant["highpass"], variant.get("hp_mode", "reflect"))
                if "interpolation" in variant:
                    ctx.set_interpolation(*variant["interpolation"])
                ctx.set_math(math)
                ctx.set_fused(mode)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=11)
                for o in range(wl.O):
                    ctx.init_templates(o, 0)
                ctx.record_moments(0)
                idx = []
                for i in range(1, T):
                    if i == 2:
                        ctx.step(i, 1.0, [i] * wl.O, normals=ev, u=us)  # (one step on host-fed draws)
                    else:
                        ctx.step(i, 1.0, [i] * wl.O, seed=11)
                    idx.append(ctx.resample_indices())
                    if mode:
                        assert ctx.last_variant()[3] & 2  # the general code
                assert (ctx.observer_status() == lib.OBS_OK).all() and (ctx.point_status() == 0).all()
                stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
                assert ("point_step" in stages) == bool(mode)
                res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), np.stack(idx)))
        np.testing.assert_array_equal(res[0][3], res[1][3])
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-12, atol=1e-13)
        assert abs(np.median(res[0][2][-1, :, 3]) - 0.15) < 0.06


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("variant", [dict(), dict(highpass=(3, 3)), dict(interpolation=(1, 1))])
def test_sixteen_bit_frames_on_the_fused_kernel(lib, variant, math):
    """uint16 frames, gray and RGB, run on the general instantiations of the fused kernel (rounds 1-3a: staged kernels
    with key histograms in memory): the pixel keys are ranked in LDS (glh_point.h: pt_tile_prep_wide) -- bit for bit the
    staged kernels, with the key tile in LDS and (mode 2) in the workspaces, host-fed and device draws; one observer
    and two.  Frames whose values span the 16-bit range, and the same scene squeezed into 300 levels (buckets without
    low bits) and a constant-free but coarse one (many equal keys per bucket)."""
    from glimpse_amd import workloads

    T = 5
    # (the last case: a 47 x 47 template -- its CDF has up to 2 209 entries, 35 KB that do not fit LDS beside the raw keys
    # at this particle count, so the stage reads it from memory and keeps the interval table over the raw keys)
    cases = (("C2", 6, 1500, 1, None, None), ("C5", 3, 2000, 3, None, None), ("C2", 4, 1200, 1, 300, None),
             ("C2", 4, 1200, 3, 40000, None), ("C3", 4, 4000, 1, None, (47, 47)))
    for name, P, N, channels, levels, tile in cases:
        if tile and variant:
            continue
        wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
        wl.channels, wl.bits = channels, 16
        if tile:
            wl.tile = tile
        frames = [wl.frames(o) for o in range(wl.O)]
        if levels:  # (a narrow range of levels, offset from zero)
            frames = [[(f.astype(np.uint32) * levels // 65535 + 1000).astype(np.uint16) for f in fo] for fo in frames]
        assert frames[0][0].dtype == np.uint16
        rng = np.random.default_rng(7)
        ev, us = rng.standard_normal((P, N, 3)), rng.random(P)
        res = []
        for mode in (1, 2, 0):
            with lib.Context(wl.P, wl.N, wl.O, max_tile=max(wl.tile), max_search_dim=160, max_frames=T) as ctx:
                workloads.setup_context(ctx, wl, frames)
                if "highpass" in variant:
                    ctx.set_highpass(variant["highpass"])
                if "interpolation" in variant:
                    ctx.set_interpolation(*variant["interpolation"])
                ctx.set_math(math)
                ctx.set_fused(mode)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=11)
                for o in range(wl.O):
                    ctx.init_templates(o, 0)
                ctx.record_moments(0)
                idx = []
                for i in range(1, T):
                    if i == 2:
                        ctx.step(i, 1.0, [i] * wl.O, normals=ev, u=us)  # (one step on host-fed draws)
                    else:
                        ctx.step(i, 1.0, [i] * wl.O, seed=11)
                    idx.append(ctx.resample_indices())
                    if mode:
                        assert ctx.last_variant()[3] & 2  # the general code
                assert (ctx.observer_status() == lib.OBS_OK).all() and (ctx.point_status() == 0).all()
                stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
                assert ("point_step" in stages) == bool(mode)
                if tile:
                    box = ctx.search_boxes()[0]
                    assert (box[:, 2] - box[:, 0]).max() > 52  # (wide enough for the raw keys to leave no room for the CDF)
                res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), np.stack(idx)))
        for other in (res[1], res[2]):
            if math == "fast" and other is res[1]:
                continue  # (mode 2 has its own bound on the per-cell sampling form: compared in exact arithmetic)
            np.testing.assert_array_equal(res[0][3], other[3])
            np.testing.assert_array_equal(res[0][0], other[0])
            np.testing.assert_array_equal(res[0][1], other[1])
            np.testing.assert_allclose(res[0][2], other[2], rtol=1e-12, atol=1e-13)
        assert abs(np.median(res[0][2][-1, :, 3]) - 0.15) < 0.06
    # a search workspace beyond 255 pixels (a tile's pixel count no longer fits a 16-bit key): the staged kernels
    with lib.Context(2, 500, 1, max_search_dim=256, max_frames=2) as ctx:
        wl = workloads.Workload("C2", n_frames=2, n_points=2, n_particles=500, imgsz=(640, 640))
        wl.bits = 16
        workloads.setup_context(ctx, wl)
        ctx.set_debug(2)
        ctx.set_frame(0)
        ctx.init_particles(seed=1)
        ctx.init_templates(0, 0)
        ctx.step(1, 1.0, [1], seed=1)
        assert "point_step" not in {k for k, v in ctx.profile_get().items() if v[1] > 0}


def test_covariances_of_the_compact_state_equal_those_of_the_expanded_state(lib):
    """glh_record_covariances reads the run-length compact state the fused step leaves through its record indices (rounds
    1-3a expanded it first, and the next frame then ran on the general instantiation): the same particles in the same
    order, hence bit for bit the covariance of the expanded state -- and the fast common instantiation keeps running."""
    from glimpse_amd import workloads

    T = 5
    wl = workloads.Workload("C2", n_frames=T, n_points=8, n_particles=2000, imgsz=(640, 640))
    frames = [wl.frames(0)]
    with lib.Context(wl.P, wl.N, 1, max_search_dim=160, max_frames=T + 1) as ctx:
        workloads.setup_context(ctx, wl, frames)
        ctx.set_math("fast")
        ctx.set_frame(0)
        ctx.init_particles(seed=4)
        ctx.init_templates(0, 0)
        ctx.record_moments(0)
        for i in range(1, T):
            ctx.step(i, 1.0, [i], seed=4)
            ctx.record_covariances(i)
            if i >= 2:
                assert ctx.last_variant()[3] == 5  # fast | contract: the state stayed compact
        compact = ctx.get_covariances(T - 1, 1)[0]
        ctx.get_particles()            # (expands the state)
        ctx.record_covariances(T)      # the same state, expanded, into another history slot
        expanded = ctx.get_covariances(T, 1)[0]
        np.testing.assert_array_equal(compact, expanded)
        assert np.isfinite(compact).all() and (np.einsum("pii->pi", compact)[:, [0, 1, 3, 4]] > 0).all()  # (z does not vary)
        mom = ctx.get_moments(T - 1, 1)[0]
        np.testing.assert_allclose(np.sqrt(np.einsum("pii->pi", compact)), mom[:, 6:], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("dtype,channels", [(np.float32, 1), (np.float64, 1), (np.float32, 3), (np.float64, 3)])
def test_float_frames_on_the_fused_kernel(lib, dtype, channels, math):
    """float32 / float64 frames, one channel and three, on the general instantiations of the fused kernel (rounds 2-3:
    staged kernels only): the point's own workgroup runs the staged tile stage (glh_kernels.h: search_tile_from_boxf --
    normalisation in the frame's dtype with NumPy's summation order, two-level ranking, median on the counts) into the
    search workspace -- bit for bit the staged kernels, host-fed and device draws, one observer and two."""
    from glimpse_amd import workloads

    T = 5
    for name, P, N in (("C2", 5, 1500), ("C5", 3, 2000), ("C3", 3, 4000)):
        wl = workloads.Workload(name, n_frames=T, n_points=P, n_particles=N, imgsz=(640, 640))
        wl.channels = channels
        # a reflectance-like image: the uint8 scene scaled into (0.1, 0.9), with a little structure below one grey level
        frames = [[(f.astype(np.float64) / 255.0 * 0.8 + 0.1 + 1e-4 * np.sin(np.arange(f.size).reshape(f.shape) * 0.37))
                   .astype(dtype) for f in wl.frames(o)] for o in range(wl.O)]
        rng = np.random.default_rng(7)
        ev, us = rng.standard_normal((P, N, 3)), rng.random(P)
        res = []
        for mode in (1, 0):
            with lib.Context(wl.P, wl.N, wl.O, max_tile=max(wl.tile), max_search_dim=160, max_frames=T) as ctx:
                for o in range(wl.O):
                    ctx.observer_init(o, T, wl.imgsz[0], wl.imgsz[1], channels, wl.sigmas[o])
                    ctx.observer_set_depth(o, dtype)
                    ctx.observer_set_cameras(o, np.tile(wl.cams[o], (T, 1)))
                    for t in range(T):
                        ctx.observer_upload_frame(o, t, frames[o][t])
                ctx.begin_sequence(wl.P, wl.N, wl.tile)
                ctx.set_motion_cartesian(wl.params)
                ctx.set_math(math)
                ctx.set_fused(mode)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=11)
                for o in range(wl.O):
                    ctx.init_templates(o, 0)
                ctx.record_moments(0)
                idx = []
                for i in range(1, T):
                    if i == 2:
                        ctx.step(i, 1.0, [i] * wl.O, normals=ev, u=us)  # (one step on host-fed draws)
                    else:
                        ctx.step(i, 1.0, [i] * wl.O, seed=11)
                    idx.append(ctx.resample_indices())
                    if mode:
                        assert ctx.last_variant()[3] & 2  # the general code
                assert (ctx.observer_status() == lib.OBS_OK).all() and (ctx.point_status() == 0).all()
                stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
                assert ("point_step" in stages) == bool(mode)
                res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), np.stack(idx)))
        np.testing.assert_array_equal(res[0][3], res[1][3])
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-12, atol=1e-13)
        assert abs(np.median(res[0][2][-1, :, 3]) - 0.15) < 0.06


@pytest.mark.parametrize("math", ["exact", "fast"])
def test_templates_up_to_63_pixels_on_the_fused_kernel(lib, math):
    """Templates beyond 48 pixels (rounds 1-3: staged kernels): the fused kernel takes sides up to 63 -- the template rows
    are padded to 64 floats, the search tile of such a template no longer fits beside it at two workgroups per CU and
    goes through the workspaces, the plan falls back to one workgroup per CU.  Bit for bit the staged kernels."""
    from glimpse_amd import workloads

    T = 4
    for tile, N in (((63, 63), 1500), ((55, 49), 3000)):
        wl = workloads.Workload("C3", n_frames=T, n_points=3, n_particles=N, imgsz=(768, 768))
        wl.tile = tile
        frames = [wl.frames(o) for o in range(wl.O)]
        res = []
        for mode in (1, 0):
            with lib.Context(wl.P, wl.N, wl.O, max_tile=max(tile), max_search_dim=192, max_frames=T) as ctx:
                workloads.setup_context(ctx, wl, frames)
                ctx.set_math(math)
                ctx.set_fused(mode)
                ctx.set_debug(2)
                ctx.set_frame(0)
                ctx.init_particles(seed=5)
                ctx.init_templates(0, 0)
                ctx.record_moments(0)
                idx = []
                for i in range(1, T):
                    ctx.step(i, 1.0, [i], seed=5)
                    idx.append(ctx.resample_indices())
                assert (ctx.observer_status() == lib.OBS_OK).all() and (ctx.point_status() == 0).all()
                stages = {k for k, v in ctx.profile_get().items() if v[1] > 0}
                assert ("point_step" in stages) == bool(mode)
                res.append((ctx.get_particles(), ctx.get_weights(), ctx.get_moments(0, T), np.stack(idx)))
        np.testing.assert_array_equal(res[0][3], res[1][3])
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("channels", [1, 3])
def test_search_tile_that_touches_the_last_pixel_of_the_frame(lib, channels):
    """A tracked point in the bottom-right corner: its search box ends at (width, height), so the tile fetch reads the frame's
    very last pixel -- which the RGB fetch (one unaligned 4-byte load per pixel) takes from the frame's last four bytes.
    Bit for bit the staged kernels."""
    from scipy.optimize import fsolve

    from glimpse_amd import synth, workloads

    T, P, N = 4, 2, 800
    W = H = 192
    wl = workloads.Workload("C2", n_frames=T, n_points=P, n_particles=N, imgsz=(W, H))
    wl.channels = channels
    cam = wl.cams[0]
    half = wl.tile[0] / 2 + 1.5  # (the box is the cloud widened by half a template and, being narrower than the template + 3, by 1.5)
    target = np.array([W - half - 0.35, H - half - 0.35])  # box: ceil(max u + half) = W for a cloud a few hundredths of a pixel wide
    xy = fsolve(lambda q: synth.project(cam, np.array([[q[0], q[1], 0.0]]))[0] - target, x0=(1.0, 1.0), xtol=1e-12)
    rng = np.random.default_rng(2)
    shape = (H, W) if channels == 1 else (H, W, 3)
    still = rng.integers(0, 256, shape, dtype=np.uint8)  # (a static texture: the point does not move)
    still[-1, -1] = 255                                   # (... whose last pixel is not like its neighbours)
    frames = [[still.copy() for _ in range(T)]]
    params = workloads.motion_params(np.array([xy, xy + (-6.0, 6.0)]))
    params[:, 2:4] = 1e-4
    params[:, 4:7] = 0.0
    params[:, 7:10] = (1e-4, 1e-4, 0.0)
    params[:, 13:16] = (1e-5, 1e-5, 0.0)
    out = []
    for mode in (1, 0):
        with lib.Context(P, N, 1, max_search_dim=96, max_frames=T) as ctx:
            workloads.setup_context(ctx, wl, frames)
            ctx.set_motion_cartesian(params)
            ctx.set_fused(mode)
            ctx.set_frame(0)
            ctx.init_particles(seed=3)
            ctx.init_templates(0, 0)
            ctx.record_moments(0)
            boxes = []
            for i in range(1, T):
                ctx.step(i, 1.0, [i], seed=3)
                boxes.append(ctx.search_boxes().copy())
            assert (ctx.observer_status() == lib.OBS_OK).all() and (ctx.point_status() == 0).all()
            out.append(dict(p=ctx.get_particles(), w=ctx.get_weights(), m=ctx.get_moments(0, T), boxes=np.stack(boxes)))
    fused, staged = out
    assert (fused["boxes"][:, 0, 0, 2:] == (W, H)).all(), fused["boxes"][:, 0, 0]  # (point 0's box ends at the frame's corner)
    np.testing.assert_array_equal(fused["boxes"], staged["boxes"])
    np.testing.assert_array_equal(fused["p"], staged["p"])
    np.testing.assert_array_equal(fused["w"], staged["w"])
    np.testing.assert_allclose(fused["m"], staged["m"], rtol=1e-12, atol=1e-13)
