"""GPU box: (a) the ground table's build-time check (interpolation error; real rays through table vs full path) for the views of the
tests, (b) per-PIXEL relative error of the whole frame, HIP vs oracle, one iteration of configs 2, 3 and 5 at spec."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_binding
lib = pkg.load_library()
lib.vpt_test_get_dir_table_check.argtypes = [C.c_void_p, C.POINTER(C.c_float * 8)]
from vpt_amd.abi import Float3

def check(hb):
    o = (C.c_float * 8)()
    lib.vpt_test_get_dir_table_check(hb.ctx.h, C.byref(o))
    return "interp %.2e | real rays (centre) %d: max %.2e, above 1e-3: %.3f %% | all variants: max %.2e, largest share above 1e-3 %.3f %% | in use %d, variants %d" % (
        o[0], int(o[2]), o[1], 100.0 * o[3] / max(1.0, o[2]), o[6], 100.0 * o[7], int(o[4]), int(o[5]))

def view(name):
    sd = pkg.scene.dragon_scene(160, 90, "c2")
    if name == "low sun": sd.kp.elevation = 3.0
    if name == "sunset": sd.kp.elevation = -1.0
    if name == "20 km up": sd.camera.origin.y += 20000.0
    if name == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, 160.0 / 90.0, 0.0)
    if name == "open lens":
        sd.camera, _, _ = pkg.scene.frame_camera(lib, [sd.volumes[0][0]], 160, 90, aperture=2.0)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    return sd

for name in ("default", "low sun", "sunset", "20 km up", "horizon in view", "open lens"):
    hb = pkg.scene.HipBinding(view(name), device=0)
    hb.render(1); hb.sync()
    print("%-16s %s" % (name, check(hb)), flush=True)
    hb.ctx.close()

def per_pixel(tag, sd):
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(1); hb.sync()
    ob = oracle_binding.OracleBinding(sd)
    ob.render(1)
    a = hb.accum.cpu().numpy().astype(np.float64); b = ob.accum.astype(np.float64)
    lum = b.max(1)
    m = lum > 1e-3
    rel = np.abs(a - b).max(1)[m] / lum[m]
    l2 = np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())
    q = np.quantile(rel, [0.5, 0.99, 0.999, 0.9999])
    print("%s: rel L2 %.2e | per pixel (lum > 1e-3: %d px): median %.1e p99 %.1e p99.9 %.1e p99.99 %.1e max %.1e | > 2e-3: %d px | %s" % (
        tag, l2, m.sum(), q[0], q[1], q[2], q[3], rel.max(), int((rel > 2e-3).sum()), check(hb)), flush=True)
    hb.ctx.close()

sd = pkg.scene.dragon_scene(1920, 1080, "c2"); pkg.atmosphere.attach_default_atmosphere(sd, device=0); per_pixel("c2", sd)
sd = pkg.scene.fireball_scene(1920, 1080, n=256, sky=True); pkg.atmosphere.attach_default_atmosphere(sd, device=0); per_pixel("c3", sd)
sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True); pkg.atmosphere.attach_default_atmosphere(sd, device=0); per_pixel("c5", sd)
for k in ("VPT_NO_DIR_TABLE",):
    os.environ[k] = "1"
sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True); pkg.atmosphere.attach_default_atmosphere(sd, device=0); per_pixel("c5 no ground table", sd)
sd = pkg.scene.dragon_scene(1920, 1080, "c2"); pkg.atmosphere.attach_default_atmosphere(sd, device=0); per_pixel("c2 no ground table", sd)
