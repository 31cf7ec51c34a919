// TEST INFRASTRUCTURE -- oracle/_ref/ref_atmosphere_model: the reference's OWN sky-model set-up, compiled for the CPU from the
// lines where they lie in /root/reference/source/atmosphere/atmosphere.cpp:
//     :123-246     cie_color_matching_function_table_value, coeff, sky_sun_radiance_to_luminance, interpolate,
//                  compute_spectral_radiance_to_luminance_factors, convert_spectrum_to_linear_srgb, adjust_units
//     :698-784     update_model(lambdas)
//     :1198-1224   the spectra loop and the model constants of init()
// over the reference's own atmosphere.h / definitions.h / constants.h.  atmosphere.cpp as a whole needs the CUDA driver API,
// tinyexr and the logger; these ranges are plain C++.  oracle/Makefile extracts them with `sed -n` into oracle/_ref/*.inc
// (git-ignored, deleted after the build); this file contains no reference code.  The class's constructor / destructor (CUDA
// allocations in the reference, :1316-1339) are given empty bodies here; the three defaults the constructor sets come from
// the command line.  It pins vpt_atmosphere_model (csrc/vpt_atmosphere.hip): tests/test_atmosphere_model.py.
//
//   ref_atmosphere_model <const_solar 0|1> <ozone 0|1> <white_balance 0|1> <use_luminance 0|1|2> <exposure> <lr> <lg> <lb> <out.bin>
//   out.bin: the scalar members of AtmosphereParameters as 32-bit floats / ints, in the order written below
#include "cuda_runtime.h"
#define private public
#include "atmosphere/atmosphere.h"      // the reference's
#undef private
// atmosphere.cpp:51 includes the reference's helper_math.h, whose `#define M_PI 3.14159265358979323846f` is the M_PI its
// code sees (no _USE_MATH_DEFINES in that file: <cmath> supplies none on the reference's platform).  Included here after
// glibc's <math.h> for the same effect: update_model's mu_s_min = cos(120 / 180 * M_PI) is evaluated with the FLOAT pi.
#include "helper_math.h"

atmosphere::atmosphere() {}
atmosphere::~atmosphere() {}

#include "atmosphere_model_functions.inc"      // atmosphere.cpp:123-246
#include "atmosphere_update_model.inc"         // atmosphere.cpp:698-784

struct Probe : atmosphere {
    void build_spectra() {
#include "atmosphere_init_spectra.inc"         // atmosphere.cpp:1198-1224
    }
    void factors_and_update(float3 lambdas) {
        // atmosphere::precompute :900-912
        if (m_use_luminance == PRECOMPUTED) sky_k_r = sky_k_g = sky_k_b = MAX_LUMINOUS_EFFICACY;
        else compute_spectral_radiance_to_luminance_factors(m_wave_lengths, m_solar_irradiance, -3, sky_k_r, sky_k_g, sky_k_b);
        compute_spectral_radiance_to_luminance_factors(m_wave_lengths, m_solar_irradiance, 0, sun_k_r, sun_k_g, sun_k_b);
        update_model(lambdas);
    }
};

static void put3(FILE* f, float3 v) { fwrite(&v, 4, 3, f); }
static void putp(FILE* f, const DensityProfile& d) {
    for (int i = 0; i < 2; ++i) {
        const float v[5] = {(float)d.layers[i].width, (float)d.layers[i].exp_term, (float)d.layers[i].exp_scale, (float)d.layers[i].linear_term, (float)d.layers[i].const_term};
        fwrite(v, 4, 5, f);
    }
}

int main(int argc, char** argv) {
    if (argc != 10) {
        fprintf(stderr, "usage: ref_atmosphere_model <const_solar> <ozone> <white_balance> <use_luminance> <exposure> <lr> <lg> <lb> <out.bin>\n");
        return 2;
    }
    Probe* P = new Probe();
    P->m_use_constant_solar_spectrum = atoi(argv[1]) != 0;
    P->m_use_ozone = atoi(argv[2]) != 0;
    P->m_do_white_balance = atoi(argv[3]) != 0;
    P->m_use_luminance = atoi(argv[4]) == 2 ? PRECOMPUTED : (atoi(argv[4]) == 1 ? APPROXIMATE : NONE);
    P->m_exposure = (float)atof(argv[5]);
    P->build_spectra();
    P->factors_and_update(make_float3((float)atof(argv[6]), (float)atof(argv[7]), (float)atof(argv[8])));
    const AtmosphereParameters& a = P->atmosphere_parameters;
    FILE* f = fopen(argv[9], "wb");
    if (!f) { perror(argv[9]); return 1; }
    put3(f, a.sky_spectral_radiance_to_luminance); put3(f, a.sun_spectral_radiance_to_luminance); put3(f, a.solar_irradiance);
    const float s0[3] = {(float)a.sun_angular_radius, (float)a.bottom_radius, (float)a.top_radius};
    fwrite(s0, 4, 3, f);
    putp(f, a.rayleigh_density); put3(f, a.rayleigh_scattering);
    putp(f, a.mie_density); put3(f, a.mie_scattering); put3(f, a.mie_extinction);
    const float g = (float)a.mie_phase_function_g;
    fwrite(&g, 4, 1, f);
    putp(f, a.absorption_density); put3(f, a.absorption_extinction); put3(f, a.ground_albedo);
    const float mu = (float)a.mu_s_min, ex = (float)a.exposure;
    fwrite(&mu, 4, 1, f);
    const int lum = (int)a.use_luminance;
    fwrite(&lum, 4, 1, f);
    put3(f, a.white_point);
    fwrite(&ex, 4, 1, f);
    fclose(f);
    return 0;
}
