"""GPU box: the ground table's build-time gate on a few views -- worst ray, unflipped rays above 1e-3, flipped rays and what they cost on average
(csrc/vpt_tail.hip: sky_dir_table_rays_kernel / _verdict_kernel).  python tools/gate_probe.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
from vpt_amd.abi import Float3
lib = pkg.load_library()
lib.vpt_test_get_dir_table_check.argtypes = [C.c_void_p, C.POINTER(C.c_float * 8)]
lib.vpt_test_get_dir_table_flips.argtypes = [C.c_void_p, C.POINTER(C.c_float * 4)]
def report(name, sd, n=2):
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(n); hb.sync()
    chk = (C.c_float * 8)(); fl = (C.c_float * 4)()
    lib.vpt_test_get_dir_table_check(hb.ctx.h, C.byref(chk)); lib.vpt_test_get_dir_table_flips(hb.ctx.h, C.byref(fl))
    print("%-18s interp err %.1e | rays %6d worst %.2e unflipped>1e-3 %.4f | flipped share %.4f mean cost %.2e | all variants: worst %.2e unflipped>1e-3 %.4f flipped %.4f cost %.2e | in use %d of variants %d"
          % (name, chk[0], int(chk[2]), chk[1], chk[3] / max(chk[2], 1.0), fl[0], fl[2], chk[6], chk[7], fl[1], fl[3], int(chk[4]), int(chk[5])))
    hb.ctx.close()
for view in ("c2", "low sun", "sunset", "20 km up", "horizon in view"):
    sd = pkg.scene.dragon_scene(160, 90, "c2")
    if view == "low sun": sd.kp.elevation = 3.0
    if view == "sunset": sd.kp.elevation = -1.0
    if view == "20 km up": sd.camera.origin.y += 20000.0
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, 160.0 / 90.0, 0.0)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    report(view, sd)
sd = pkg.scene.dragon_scene(1920, 1080, "c2"); pkg.atmosphere.attach_default_atmosphere(sd, device=0); report("c2 1080p", sd)
sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True); pkg.atmosphere.attach_default_atmosphere(sd, device=0); report("c5 4K open lens", sd)
