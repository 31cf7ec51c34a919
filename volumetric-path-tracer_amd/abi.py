"""ctypes mirror of include/vpt_abi.h (field-for-field; ctypes applies the platform's
natural alignment, so the layouts match what hipcc/g++ produce for the C structs).

tests/test_abi_layout.py checks every sizeof/offsetof against the compiled library.
"""
import ctypes as C

vpt_texture_t = C.c_ulonglong


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Int3(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("z", C.c_int)]


class UInt2(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint)]


class Camera(C.Structure):
    _fields_ = [
        ("time1", C.c_float), ("time0", C.c_float),
        ("origin", Float3), ("focus_dist", C.c_float),
        ("lower_left_corner", Float3), ("horizontal", Float3), ("vertical", Float3),
        ("u", Float3), ("v", Float3), ("w", Float3),
        ("lens_radius", C.c_float), ("viz_dof", C.c_ubyte),
    ]


class PointLight(C.Structure):
    _fields_ = [("pos", Float3), ("dir", Float3), ("power", C.c_float), ("color", Float3)]


class LightList(C.Structure):
    _fields_ = [("num_lights", C.c_uint), ("light_ptr", C.POINTER(PointLight))]


class Sphere(C.Structure):
    _fields_ = [("center", Float3), ("radius", C.c_float), ("color", Float3), ("roughness", C.c_float)]


class VdbInfo(C.Structure):
    _fields_ = [
        ("voxelsize", C.c_float), ("dim", Int3), ("bmin", Float3), ("bmax", Float3),
        ("max_density", C.c_float), ("min_density", C.c_float),
        ("has_color", C.c_ubyte), ("has_emission", C.c_ubyte), ("matte", C.c_ubyte),
        ("density_texture", vpt_texture_t), ("emission_texture", vpt_texture_t), ("color_texture", vpt_texture_t),
    ]


class GpuVdb(C.Structure):
    _fields_ = [("vdb_info", VdbInfo), ("xform", (C.c_float * 4) * 4)]


class DensityProfileLayer(C.Structure):
    _fields_ = [("width", C.c_float), ("exp_term", C.c_float), ("exp_scale", C.c_float),
                ("linear_term", C.c_float), ("const_term", C.c_float)]


class DensityProfile(C.Structure):
    _fields_ = [("layers", DensityProfileLayer * 2)]


class AtmosphereParameters(C.Structure):
    _fields_ = [
        ("sky_spectral_radiance_to_luminance", Float3),
        ("sun_spectral_radiance_to_luminance", Float3),
        ("solar_irradiance", Float3),
        ("angle", C.c_float), ("bottom_radius", C.c_float), ("top_radius", C.c_float),
        ("use_luminance", C.c_int),
        ("rayleigh_density", DensityProfile), ("rayleigh_scattering", Float3),
        ("mie_density", DensityProfile), ("mie_scattering", Float3), ("mie_extinction", Float3),
        ("mie_phase_function_g", C.c_float),
        ("absorption_density", DensityProfile), ("absorption_extinction", Float3),
        ("ground_albedo", Float3), ("sun_angular_radius", C.c_float), ("mu_s_min", C.c_float),
        ("exposure", C.c_float), ("white_point", Float3),
        ("delta_irradience_buffer", C.c_void_p),
        ("delta_rayleigh_scattering_buffer", C.c_void_p),
        ("delta_mie_scattering_buffer", C.c_void_p),
        ("delta_scattering_density_buffer", C.c_void_p),
        ("delta_multiple_scattering_buffer", C.c_void_p),
        ("transmittance_buffer", C.c_void_p),
        ("irradiance_buffer", C.c_void_p),
        ("scattering_buffer", C.c_void_p),
        ("optional_mie_single_scattering_buffer", C.c_void_p),
        ("transmittance_texture", vpt_texture_t),
        ("scattering_texture", vpt_texture_t),
        ("irradiance_texture", vpt_texture_t),
        ("single_mie_scattering_texture", vpt_texture_t),
    ]


class KernelParams(C.Structure):
    _fields_ = [
        ("render", C.c_ubyte), ("debug", C.c_ubyte),
        ("resolution", UInt2), ("exposure_scale", C.c_float),
        ("display_buffer", C.c_void_p), ("raw_buffer", C.c_void_p), ("blue_noise_buffer", C.c_void_p),
        ("emission_texture", C.c_void_p), ("emission_scale", C.c_float), ("emission_pivot", C.c_float),
        ("density_color_texture", C.c_void_p),
        ("iteration", C.c_uint), ("accum_buffer", C.c_void_p), ("depth_buffer", C.c_void_p),
        ("max_interactions", C.c_uint), ("ray_depth", C.c_int), ("volume_depth", C.c_int),
        ("min_extinction", C.c_float), ("phase_g1", C.c_float), ("phase_g2", C.c_float), ("phase_f", C.c_float),
        ("albedo", Float3), ("extinction", Float3), ("transmittance", Float3),
        ("tr_depth", C.c_float), ("density_mult", C.c_float),
        ("environment_type", C.c_uint), ("azimuth", C.c_float), ("elevation", C.c_float),
        ("sun_color", Float3), ("sky_color", Float3), ("sun_mult", C.c_float), ("sky_mult", C.c_float),
        ("energy_inject", C.c_double),
        ("env_tex", vpt_texture_t), ("env_sample_tex_res", C.c_int),
        ("sky_tex", vpt_texture_t), ("env_func_tex", vpt_texture_t), ("env_cdf_tex", vpt_texture_t),
        ("env_marginal_func_tex", vpt_texture_t), ("env_marginal_cdf_tex", vpt_texture_t),
        ("env_marginal_int", C.c_float),
        ("debug_buffer", C.c_void_p), ("cost_buffer", C.c_void_p),
        ("integrator", C.c_int),
    ]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("depth", C.c_int), ("channels", C.c_int),
                ("normalized_coords", C.c_int), ("filter_mode", C.c_int), ("address_mode", C.c_int * 3)]


class RenderStats(C.Structure):
    _fields_ = [("samples", C.c_ulonglong), ("density_lookups", C.c_ulonglong), ("color_lookups", C.c_ulonglong),
                ("emission_lookups", C.c_ulonglong), ("tracking_steps", C.c_ulonglong), ("skip_steps", C.c_ulonglong), ("queued_rays", C.c_ulonglong),
                ("trace_ms", C.c_float), ("raygen_ms", C.c_float), ("tail_ms", C.c_float),
                ("density_fetches", C.c_ulonglong), ("color_fetches", C.c_ulonglong), ("emission_fetches", C.c_ulonglong),
                ("density_zero_skips", C.c_ulonglong)]


class AtmosphereModelOptions(C.Structure):
    _fields_ = [("use_constant_solar_spectrum", C.c_int), ("use_ozone", C.c_int), ("do_white_balance", C.c_int), ("use_luminance", C.c_int),
                ("half_precision", C.c_int), ("exposure", C.c_float), ("lambdas", C.c_double * 3), ("length_unit_in_meters", C.c_double)]


COMM_ID_BYTES = 128
ADDR_WRAP, ADDR_CLAMP = 0, 1
FILTER_POINT, FILTER_LINEAR = 0, 1

E_NAMES = {0: "VPT_OK", -1: "VPT_E_INVALID", -2: "VPT_E_NO_DEVICE", -3: "VPT_E_HIP", -4: "VPT_E_NOMEM",
           -5: "VPT_E_NOT_READY", -6: "VPT_E_UNSUPPORTED", -7: "VPT_E_IO"}
