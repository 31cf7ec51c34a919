"""Diagnostic for tests/test_gpu_parity.py::test_random_setups_vs_oracle: one of its set-ups (seed, case) rendered with each sky cache switched off in turn, against the oracle.
   python tools/sky_random_probe.py --seed 1 --case 2"""
import argparse, os, sys, ctypes as C
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import __graft_entry__ as ge
ge.build(); ge.build_oracle()
pkg = ge.load_package()
import oracle_binding
from vpt_amd.abi import Float3

ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--case", type=int, default=2); ap.add_argument("--spp", type=int, default=3)
a = ap.parse_args()
lib = pkg.load_library()

def rel_l2(x, y):
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    return float(np.sqrt(((x - y) ** 2).sum()) / max(1e-30, np.sqrt((y ** 2).sum())))

from random_setups import dragon_setup
rs = np.random.RandomState(500 + a.seed)
for case in range(8):
    sd, w, h, sky, desc = dragon_setup(pkg, rs, case)
    if case != a.case: continue
    print("case %d: %s" % (case, desc))
    ob = oracle_binding.OracleBinding(sd); ob.render(a.spp, nthreads=os.cpu_count() or 1)
    ref = ob.accum.reshape(h, w, 3)
    for sw in (None, "VPT_NO_SKY_PATCH", "VPT_NO_SKY_DOME", "VPT_NO_DIR_TABLE", "VPT_NO_CAM_TABLE", "VPT_NO_PIXEL_CULL", "VPT_NO_LEAN_TAIL", "VPT_NO_HEADS", "ALL"):
        names = ["VPT_NO_SKY_PATCH", "VPT_NO_SKY_DOME", "VPT_NO_DIR_TABLE", "VPT_NO_CAM_TABLE", "VPT_NO_LEAN_TAIL"] if sw == "ALL" else ([sw] if sw else [])
        for n in names: os.environ[n] = "1"
        hb = pkg.scene.HipBinding(sd, device=0); hb.render(a.spp); hb.sync()
        got = hb.accum.cpu().numpy().reshape(h, w, 3); hb.ctx.close()
        for n in names: del os.environ[n]
        err = np.abs(got - ref).sum(-1); rel = err / np.maximum(1e-6, np.abs(ref).sum(-1))
        yx = np.unravel_index(np.argmax(err), err.shape)
        print("%-20s rel L2 %.3e | pixels with rel err > 1e-3: %5d of %d, > 1e-2: %4d | worst at (x %d, y %d): got %s ref %s" % (sw or "default", rel_l2(got, ref), int((rel > 1e-3).sum()), w * h, int((rel > 1e-2).sum()), yx[1], yx[0], got[yx], ref[yx]))
        if sw is None:
            rows = (rel > 1e-3).sum(1); print("  rows with the most bad pixels:", np.argsort(-rows)[:6], rows[np.argsort(-rows)[:6]])
    # the worst pixel of the default render, sample by sample
    hb = pkg.scene.HipBinding(sd, device=0); hb.render(a.spp); hb.sync()
    got = hb.accum.cpu().numpy().reshape(h, w, 3); hb.ctx.close()
    err = np.abs(got - ref).sum(-1)
    for (py, px) in [np.unravel_index(i, err.shape) for i in np.argsort(-err.ravel())[:3]]:
        for k in range(a.spp):
            hb = pkg.scene.HipBinding(sd, device=0); hb.render(1, iteration=k); hb.sync()
            v = hb.accum.cpu().numpy().reshape(h, w, 3)[py, px] * np.float32(k + 1); dd = hb.depth.cpu().numpy().reshape(h, w)[py, px] * np.float32(k + 1); hb.ctx.close()
            o = ob.sample_pixel(int(px), int(py), iteration=k)
            print("pixel (x %d, y %d) iteration %d: HIP value %s depth*%d %.6f | oracle {value, tr, depth} %s" % (px, py, k, v, k + 1, dd, o))
    hb = pkg.scene.HipBinding(sd, device=0); hb.ctx.set_counting(True); hb.render(a.spp); hb.sync(); st = hb.ctx.stats()
    gc = hb.accum.cpu().numpy().reshape(h, w, 3); hb.ctx.close()
    print("counting render: rel L2 vs oracle %.3e, vs timed %.3e; counts HIP %s oracle %s" % (rel_l2(gc, ref), rel_l2(gc, got), (st.density_lookups, st.tracking_steps, st.skip_steps), (ob.stats.density_lookups, ob.stats.tracking_steps, ob.stats.skip_steps)))
