import sys
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package()
for env in (0, 1):
    sd = pkg.scene.dragon_scene(1920, 1080, "c2")
    sd.kp.integrator = 1
    if env == 0:
        sd.env_cdf = pkg.host.env_cdf_build(sd.kp)
    else:
        sd.kp.environment_type = 1
        sd.env_map = pkg.scene.hdri_map(2048, 1024)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(4); hb.sync()
    hb.render(8, iteration=0); hb.sync()
    st = hb.ctx.stats()
    n = 1920 * 1080 * 8
    tot = st.raygen_ms + st.trace_ms + st.tail_ms
    print("vol_integrator on dragon, environment_type %d: raygen %.2f trace %.2f tail %.2f ms per 8 spp -> %.0f Msamples/s" % (env, st.raygen_ms, st.trace_ms, st.tail_ms, n / tot / 1e3))
