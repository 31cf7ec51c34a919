"""GPU box: config 5 at spec (4K, 100 instances, open lens, sun + sky): HIP vs oracle, per image and per pixel, with and without the
per-frame sky tables.  python tools/c5_parity_probe.py [iterations]   (VPT_LIB_PATH selects a study library)"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_binding
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / np.sqrt((b ** 2).sum()))
def per_pixel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    lum = b.max(1); m = lum > 1e-3
    return np.abs(a - b).max(1)[m] / lum[m]
sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True)
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
ob = oracle_binding.OracleBinding(sd)
ob.render(n_it)
res = {}
for name, env in (("tables", {}), ("no ground table", {"VPT_NO_DIR_TABLE": "1"})):
    for k in ("VPT_NO_DIR_TABLE", "VPT_NO_CAM_TABLE"): os.environ.pop(k, None)
    os.environ.update(env)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(n_it); hb.sync()
    res[name] = hb.accum.cpu().numpy()
    q = np.quantile(per_pixel(res[name], ob.accum), [0.5, 0.99, 0.999, 1.0])
    chk = (C.c_float * 8)()
    lib = pkg.load_library()
    lib.vpt_test_get_dir_table_check.argtypes = [C.c_void_p, C.POINTER(C.c_float * 8)]
    lib.vpt_test_get_dir_table_check(hb.ctx.h, C.byref(chk))
    print("%-16s %d iterations: rel L2 vs oracle %.3e | per pixel median %.1e p99 %.2e p99.9 %.2e worst %.2e | depth identical %s | ground-table variants in use %d"
          % (name, n_it, rel_l2(res[name], ob.accum), q[0], q[1], q[2], q[3], np.array_equal(hb.depth.cpu().numpy(), ob.depth), int(chk[5])))
    hb.ctx.close()
