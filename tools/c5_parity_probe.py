"""GPU box: config 5 at spec (4K, 100 instances, open lens, sun + sky), one iteration: HIP vs oracle with and without the per-frame sky tables."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_binding
def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / np.sqrt((b ** 2).sum()))
sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True)
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
ob = oracle_binding.OracleBinding(sd)
ob.render(1)
res = {}
for name, env in (("tables", {}), ("no ground table", {"VPT_NO_DIR_TABLE": "1"}), ("no tables", {"VPT_NO_DIR_TABLE": "1", "VPT_NO_CAM_TABLE": "1"})):
    for k in ("VPT_NO_DIR_TABLE", "VPT_NO_CAM_TABLE"): os.environ.pop(k, None)
    os.environ.update(env)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(1); hb.sync()
    res[name] = hb.accum.cpu().numpy()
    print("%-16s rel L2 vs oracle %.3e   depth identical %s" % (name, rel_l2(res[name], ob.accum), np.array_equal(hb.depth.cpu().numpy(), ob.depth)))
    hb.ctx.close()
print("tables vs no tables: %.3e" % rel_l2(res["tables"], res["no tables"]))
