"""GPU box: the view-point ground table of the environment tail (vpt_sky.h) against the full evaluation of every ground hit.
    python tools/dir_table_probe.py [W H spp]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
W, H, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (480, 270, 8)
lib = pkg.load_library()
NN = int(os.environ.get('DT_NN', '64'))
lib.vpt_test_get_dir_table_error.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint)]
for name, tweak in (("c2", None), ("c2 low sun", lambda sd: setattr(sd.kp, "elevation", 3.0)), ("c2 sunset", lambda sd: setattr(sd.kp, "elevation", -1.0)),
                    ("c2 high camera", "high"), ("c2 aperture 2", "lens")):
    imgs = {}
    for mode in ("table", "full"):
        if mode == "full": os.environ["VPT_NO_DIR_TABLE"] = "1"
        else: os.environ.pop("VPT_NO_DIR_TABLE", None)
        sd = pkg.scene.dragon_scene(W, H, "c2")
        if callable(tweak): tweak(sd)
        if tweak == "high":
            sd.camera.origin.y += 20000.0
        if tweak == "lens":
            sd.camera, _, _ = pkg.scene.frame_camera(lib, [sd.volumes[0][0]], W, H, aperture=2.0)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.render(spp); hb.sync()
        imgs[mode] = hb.accum.cpu().numpy().reshape(H, W, 3).astype(np.float64)
        if mode == "table":
            b, e, cell = C.c_int(0), C.c_float(0), C.c_uint(0)
            lib.vpt_test_get_dir_table_error(hb.ctx.h, C.byref(b), C.byref(e), C.byref(cell))
    a, f = imgs["table"], imgs["full"]
    diff = np.abs(a - f).max(axis=2)
    rel = diff / np.maximum(f.max(axis=2), 1e-6)
    print("%-16s table built %d  mid-cell error %.3e at (d %d, nu %d) | pixels differing %6d of %d  max rel diff %.3e  rel L2 %.3e  mean image %.4f" % (
        name, b.value, e.value, cell.value // (NN - 1), cell.value % (NN - 1), int((diff > 0).sum()), W * H, rel.max(), np.sqrt(((a - f) ** 2).sum() / (f ** 2).sum()), f.mean()))
