import sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
sd = pkg.scene.dragon_scene(1920, 1080, "c2")
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
hb = pkg.scene.HipBinding(sd, device=0)
hb.render(4); hb.sync()
for n in (1, 4, 64):
    hb.kp.iteration = 0
    t = time.perf_counter()
    for i in range(64 // n):
        hb.render(n)
    hb.sync()
    dt = time.perf_counter() - t
    print("64 iterations as %2d launches of %2d: %.2f ms -> %.0f Msamples/s" % (64 // n, n, dt * 1e3, 1920 * 1080 * 64 / dt / 1e6))
