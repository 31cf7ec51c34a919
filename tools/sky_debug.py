import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_binding
sd = pkg.scene.dragon_scene(160, 90, "c2")
sd.kp.sun_mult = 0.0
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
hb = pkg.scene.HipBinding(sd, device=0); hb.render(1); hb.sync()
ob = oracle_binding.OracleBinding(sd); ob.render(1)
g = hb.accum.cpu().numpy().reshape(90, 160, 3).astype(np.float64); o = ob.accum.reshape(90, 160, 3).astype(np.float64)
d = g - o
print(os.environ.get("VPT_LIB_PATH", "default"), "rel L2 %.3e" % (np.sqrt((d**2).sum()) / np.sqrt((o**2).sum())), "max abs %.3e" % np.abs(d).max(), "mean", o.mean())
rows = np.sqrt((d**2).sum((1, 2)) / (o**2).sum((1, 2)))
print(" per-row rel L2 (every 10th):", np.array2string(rows[::10], precision=2))
idx = np.argsort(-np.abs(d).max(-1).ravel())[:5]
for i in idx:
    y, x = divmod(i, 160); print("  px", x, y, "hip", g[y, x], "orc", o[y, x])
rel = np.abs(d).max(-1) / np.maximum(o.max(-1), 1e-9)
print(" fraction of pixels with rel err > 1e-3: %.4f, > 1e-4: %.4f; median rel err %.3e" % ((rel > 1e-3).mean(), (rel > 1e-4).mean(), np.median(rel)))
for (x, y) in ((5, 5), (155, 85), (80, 45), (10, 80)):
    print("  px", x, y, "hip", g[y, x], "orc", o[y, x], "depth", float(hb.depth.cpu().numpy().reshape(90, 160)[y, x]))
import hashlib
print(" accum md5", hashlib.md5(hb.accum.cpu().numpy().tobytes()).hexdigest(), "lib", pkg.load_library()._name)
