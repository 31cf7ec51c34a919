"""GPU box: ground table vs full evaluation per DIRECTION (vpt_test_sky_samples) over the lower hemisphere of config 2's view point."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
lib = pkg.load_library()
lib.vpt_test_sky_samples.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
sd = pkg.scene.dragon_scene(64, 36, "c2")
LENS = float(os.environ.get("LENS", "0"))                 # aperture: origins spread over a disc of that diameter around the camera
if len(sys.argv) > 1: sd.camera.origin.y += float(sys.argv[1])
if LENS: sd.camera, _, _ = pkg.scene.frame_camera(lib, [sd.volumes[0][0]], 64, 36, aperture=LENS)
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
hb = pkg.scene.HipBinding(sd, device=0)
hb.render(1); hb.sync()
rng = np.random.default_rng(5)
n = 1 << 20
# elevation below the horizontal: log-uniform from 1e-5 rad to pi/2, azimuth uniform
el = -np.exp(rng.uniform(np.log(1e-5), np.log(np.pi / 2), n))
az = rng.uniform(0, 2 * np.pi, n)
d = np.stack([np.cos(el) * np.cos(az), np.sin(el), np.cos(el) * np.sin(az)], 1).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
org = None
if LENS:
    c = sd.camera
    ph, rr = rng.uniform(0, 2 * np.pi, n), 0.5 * LENS * np.sqrt(rng.uniform(0, 1, n))
    U = np.array([c.u.x, c.u.y, c.u.z]); V = np.array([c.v.x, c.v.y, c.v.z])
    org = (np.array([c.origin.x, c.origin.y, c.origin.z])[None, :] + (rr * np.cos(ph))[:, None] * U + (rr * np.sin(ph))[:, None] * V).astype(np.float32)
out = {}
for use in (1, 0):
    o = np.zeros((n, 3), np.float32)
    rc = lib.vpt_test_sky_samples(hb.ctx.h, n, org.ctypes.data if org is not None else None, d.ctypes.data, use, o.ctypes.data)
    assert rc == 0, rc
    out[use] = o.astype(np.float64)
rel = np.abs(out[1] - out[0]).max(1) / np.maximum(out[0].max(1), 1e-9)
print("differing %d of %d, max rel %.3e, rms rel %.3e" % ((rel > 0).sum(), n, rel.max(), np.sqrt((rel ** 2).mean())))
edges = np.log10(np.array([1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1.0, 1.5708]))
b = np.digitize(np.log10(-el), edges)
for k in range(1, len(edges)):
    m = b == k
    if m.any():
        print("elevation -%.0e..-%.0e rad: n %7d  differing %7d  max rel %.3e  p99 %.3e  median %.3e" % (10 ** edges[k - 1], 10 ** edges[k], m.sum(), (rel[m] > 0).sum(), rel[m].max(), np.quantile(rel[m], 0.99), np.median(rel[m])))
w = np.argsort(rel)[::-1][:8]
for i in w:
    print("worst: el %.6f rad az %.3f rel %.3e table %s full %s" % (el[i], az[i], rel[i], np.round(out[1][i], 5), np.round(out[0][i], 5)))
