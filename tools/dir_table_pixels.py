"""GPU box: where do the ground table and the full evaluation differ most?  (1 iteration, config 2 at full size)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
W, H = 1920, 1080
imgs = {}
for mode in ("table", "full"):
    if mode == "full": os.environ["VPT_NO_DIR_TABLE"] = "1"
    else: os.environ.pop("VPT_NO_DIR_TABLE", None)
    sd = pkg.scene.dragon_scene(W, H, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(1); hb.sync()
    imgs[mode] = hb.accum.cpu().numpy().reshape(H, W, 3).astype(np.float64)
a, f = imgs["table"], imgs["full"]
d = np.abs(a - f).max(axis=2)
rel = d / np.maximum(f.max(axis=2), 1e-6)
idx = np.argsort(rel.ravel())[::-1][:12]
for i in idx:
    y, x = divmod(int(i), W)
    print("pixel (%4d,%4d) rel %.3e table %s full %s" % (x, y, rel[y, x], np.round(a[y, x], 5), np.round(f[y, x], 5)))
print("pixels with rel diff > 1e-3: %d, > 1e-2: %d; rows of those > 1e-2:" % ((rel > 1e-3).sum(), (rel > 1e-2).sum()), np.unique(np.nonzero(rel > 1e-2)[0])[:40])
h = np.histogram(np.log10(np.maximum(rel[rel > 0], 1e-12)), bins=np.arange(-9, 1))
print("log10 rel diff histogram:", list(zip(h[1][:-1].astype(int), h[0])))
