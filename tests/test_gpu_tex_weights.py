"""DIAGNOSTIC switch VPT_TEX_WEIGHTS=fixed8 ("what is not pinned"): the reference's tex3D look-ups (render_kernel.cu:999-1014) run on the
CUDA texture unit, whose linear filter holds its weights in 9-bit fixed point with 8 fractional bits; the parity contract of this repository -- the
reference compiled for the CPU, the oracle, the HIP path -- is binary32 weights.  The switch models the hardware's quantisation on the volume-grid
look-ups in BOTH the HIP path (TraceParams::tex_fixed8, make_taps) and the oracle (orc_set_volume_tex_weights): the two must then agree exactly as
they do with binary32 weights (same decisions: depth, alpha, every count), and differ from the binary32 render -- which is the size of the question."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


COUNTS = ("density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps")


def _scenes(pkg):
    yield "dragon", pkg.scene.dragon_scene(128, 72, "sun"), 3, 5e-6
    yield "fireball", pkg.scene.fireball_scene(96, 64, n=37), 3, 5e-6                              # emission grid
    yield "instanced", pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3), 2, 5e-6       # colour grids, open lens
    sd = pkg.scene.cloud_scene(96, 64, shape=(40, 24, 32), env=(64, 32))                            # vol_integrator, HDRI (+ the value-only sky: 1e-3)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    yield "cloud_vol", sd, 2, 1e-3


@pytest.fixture()
def fixed8(orc):
    """both switches on for the test, off afterwards (the HIP one is read when a context is created)"""
    prev = orc.orc_set_volume_tex_weights(8)
    os.environ["VPT_TEX_WEIGHTS"] = "fixed8"
    yield
    os.environ.pop("VPT_TEX_WEIGHTS", None)
    orc.orc_set_volume_tex_weights(prev)


def test_fixed8_weights_hip_equals_oracle_and_differs_from_binary32(pkg, orc, fixed8):
    import oracle_binding
    report = []
    for name, sd, spp, tol in _scenes(pkg):
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(True)
        hb.render(spp); hb.sync()
        ob = oracle_binding.OracleBinding(sd)
        ob.render(spp)
        got = hb.accum.cpu().numpy()
        assert np.isfinite(got).all() and ob.accum.max() > 0
        assert rel_l2(got, ob.accum) <= tol, name
        np.testing.assert_allclose(hb.depth.cpu().numpy(), ob.depth, rtol=1e-6, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(hb.raw.cpu().numpy()[:, 3], ob.raw[:, 3], rtol=1e-6, atol=1e-6, err_msg=name)
        st = hb.ctx.stats()
        for c in COUNTS:
            assert getattr(st, c) == getattr(ob.stats, c), (name, c)
        # ... and the binary32 render of the same scene (the contract) is a different one
        os.environ.pop("VPT_TEX_WEIGHTS")
        try:
            h32 = pkg.scene.HipBinding(sd, device=0)
            h32.ctx.set_counting(True)
            h32.render(spp); h32.sync()
        finally:
            os.environ["VPT_TEX_WEIGHTS"] = "fixed8"
        s32 = h32.ctx.stats()
        d = rel_l2(got, h32.accum.cpu().numpy())
        report.append((name, d, st.tracking_steps, s32.tracking_steps))
        assert d > 0.0, name
    print("fixed8 vs binary32 weights (rel. L2 of the accumulated image, tracking steps): " +
          "; ".join("%s %.2e (%d / %d)" % r for r in report))


def test_default_is_binary32(pkg, orc):
    assert "VPT_TEX_WEIGHTS" not in os.environ
    prev = orc.orc_set_volume_tex_weights(32)
    assert prev == 32
