"""How tight is the never-traced mask?  Config 2 at 1080p: pixels raygen skips, samples it emits, rays it queues.   python tools/mask_probe.py"""
import ctypes as C, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
ge.build()
pkg = ge.load_package()
lib = pkg.load_library()
lib.vpt_test_count_never_traced.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w, h, it = 1920, 1080, 16
sd = pkg.scene.dragon_scene(w, h, "c2")
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
for sw in (None, "VPT_NO_LEAF_CULL", "VPT_NO_PIXEL_CULL"):
    if sw: os.environ[sw] = "1"
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(it); hb.sync()
    n = C.c_ulonglong(0); lib.vpt_test_count_never_traced(hb.ctx.h, C.byref(n))
    st = hb.ctx.stats()
    hb.ctx.close()
    if sw: del os.environ[sw]
    live = w * h - n.value
    print("%-18s never-traced pixels %8d (%.1f %%), live %8d (%.1f %%); rays queued per iteration %.0f = %.1f %% of the frame's samples, %.1f %% of the live ones" % (
        sw or "default", n.value, 100.0 * n.value / (w * h), live, 100.0 * live / (w * h), st.queued_rays / it, 100.0 * st.queued_rays / it / (w * h), 100.0 * st.queued_rays / it / max(1, live)))
