// TEST INFRASTRUCTURE -- oracle/_ref/libvptref_atm.so: the reference's OWN atmosphere precomputation kernels
// (/root/reference/source/atmosphere/atmosphere_kernels.cu), compiled unmodified for the CPU where they lie, over the
// stand-in CUDA headers of this directory.  It pins the product's table precompute (csrc/vpt_atmosphere.hip,
// SURVEY 8f-1): tests/test_gpu_atmosphere_vs_ref.py compares the four tables.
//
// This file contains no reference code.  It allocates the nine scratch buffers (zeroed; the reference cudaMallocs them
// and never reads one before writing it when blend is false), and launches the reference's six kernels in the order,
// with the grids and with the ARGUMENT BYTES atmosphere::precompute hands to cuLaunchKernel
// (source/atmosphere/atmosphere.cpp:888-1116).  That includes the reference's quirk that calculate_indirect_irradiance
// and calculate_multiple_scattering declare `const int blend` while the host passes the address of a float4 whose
// first lane is 0.0f: the kernels see blend == 0 (atmosphere.cpp:1052-1083).
#include "cuda_runtime.h"
thread_local uint3 blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;

#include "atmosphere/atmosphere_kernels.cu"   // the reference kernels (found through -I/root/reference/source)

#include "../../include/vpt_abi.h"
#include <atomic>
#include <functional>
#include <thread>

namespace {

float3 cv(const vpt_float3& a) { return make_float3(a.x, a.y, a.z); }

void convert_profile(DensityProfile& d, const vpt_density_profile& s) {
    for (int i = 0; i < 2; ++i) {
        d.layers[i].width = s.layers[i].width;
        d.layers[i].exp_term = s.layers[i].exp_term;
        d.layers[i].exp_scale = s.layers[i].exp_scale;
        d.layers[i].linear_term = s.layers[i].linear_term;
        d.layers[i].const_term = s.layers[i].const_term;
    }
}

// model scalars of a pass (everything but the nine buffers)
void set_scalars(AtmosphereParameters& atm, const vpt_atmosphere_parameters* vatm) {
    atm.sky_spectral_radiance_to_luminance = cv(vatm->sky_spectral_radiance_to_luminance);
    atm.sun_spectral_radiance_to_luminance = cv(vatm->sun_spectral_radiance_to_luminance);
    atm.solar_irradiance = cv(vatm->solar_irradiance);
    atm.angle = vatm->angle;
    atm.bottom_radius = vatm->bottom_radius;
    atm.top_radius = vatm->top_radius;
    atm.use_luminance = vatm->use_luminance;
    convert_profile(atm.rayleigh_density, vatm->rayleigh_density);
    atm.rayleigh_scattering = cv(vatm->rayleigh_scattering);
    convert_profile(atm.mie_density, vatm->mie_density);
    atm.mie_scattering = cv(vatm->mie_scattering);
    atm.mie_extinction = cv(vatm->mie_extinction);
    atm.mie_phase_function_g = vatm->mie_phase_function_g;
    convert_profile(atm.absorption_density, vatm->absorption_density);
    atm.absorption_extinction = cv(vatm->absorption_extinction);
    atm.ground_albedo = cv(vatm->ground_albedo);
    atm.sun_angular_radius = vatm->sun_angular_radius;
    atm.mu_s_min = vatm->mu_s_min;
    atm.exposure = vatm->exposure;
    atm.white_point = cv(vatm->white_point);
}

// one "thread" per texel, blockDim = 1: every kernel only writes its own texel and reads tables written by
// earlier launches, so any execution order inside a launch gives the same result
void launch(int nx, int ny, int nz, int threads, const std::function<void()>& kernel) {
    std::atomic<long> cursor{0};
    const long total = (long)nx * ny * nz;
    auto worker = [&]() {
        blockDim = dim3(1, 1, 1);
        gridDim = dim3((unsigned)nx, (unsigned)ny, (unsigned)nz);
        threadIdx = make_uint3(0, 0, 0);
        for (;;) {
            const long first = cursor.fetch_add(256);
            if (first >= total) break;
            const long last = first + 256 < total ? first + 256 : total;
            for (long i = first; i < last; ++i) {
                blockIdx = make_uint3((unsigned)(i % nx), (unsigned)((i / nx) % ny), (unsigned)(i / ((long)nx * ny)));
                kernel();
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
}

}  // namespace

// tables out: float4 texels, x fastest; transmittance/irradiance 256x64 and 64x16 as constants.h sizes them,
// scattering / single_mie SCATTERING_TEXTURE_WIDTH x HEIGHT x DEPTH.  A NULL output is skipped.
// max_z > 0 restricts the three scattering-sized launches of orders >= 2 to the first max_z depth slices (a cheaper,
// partial run for small machines: the slices that are computed are exact only for order 2, whose inputs are complete).
// npasses > 0: the PRECOMPUTED luminance mode of atmosphere::init (atmosphere.cpp:1237-1268): pass i runs atmosphere::precompute with
// the model scalars passes[i] (update_model(lambdas_i)), the luminance-from-radiance matrix lfrm[9 i .. 9 i + 8] and blend = i > 0;
// then compute_transmittance with the scalars `vatm`.  npasses == 0: the one ordinary pass (identity matrix, no blend).
extern "C" int ref_atmosphere_precompute_passes(const vpt_atmosphere_parameters* vatm, const vpt_atmosphere_parameters* passes, const double* lfrms,
                                                int npasses, int num_scattering_orders, int nthreads,
                                                float* transmittance, float* irradiance, float* scattering, float* single_mie,
                                                float* delta_scattering_density, int max_z, float guard_fill);
extern "C" int ref_atmosphere_precompute(const vpt_atmosphere_parameters* vatm, int num_scattering_orders, int nthreads,
                                         float* transmittance, float* irradiance, float* scattering, float* single_mie,
                                         float* delta_scattering_density, int max_z, float guard_fill) {
    return ref_atmosphere_precompute_passes(vatm, nullptr, nullptr, 0, num_scattering_orders, nthreads, transmittance, irradiance, scattering,
                                            single_mie, delta_scattering_density, max_z, guard_fill);
}
extern "C" int ref_atmosphere_precompute_passes(const vpt_atmosphere_parameters* vatm, const vpt_atmosphere_parameters* passes, const double* lfrms,
                                                int npasses, int num_scattering_orders, int nthreads,
                                                float* transmittance, float* irradiance, float* scattering, float* single_mie,
                                                float* delta_scattering_density, int max_z, float guard_fill) {
    if (!vatm) return -1;
    if (num_scattering_orders < 1) num_scattering_orders = 4;
    const int threads = nthreads > 1 ? nthreads : 1;
    const size_t nt = (size_t)TRANSMITTANCE_TEXTURE_WIDTH * TRANSMITTANCE_TEXTURE_HEIGHT;
    const size_t ni = (size_t)IRRADIANCE_TEXTURE_WIDTH * IRRADIANCE_TEXTURE_HEIGHT;
    const size_t ns = (size_t)SCATTERING_TEXTURE_WIDTH * SCATTERING_TEXTURE_HEIGHT * SCATTERING_TEXTURE_DEPTH;

    AtmosphereParameters atm;
    std::memset(&atm, 0, sizeof(atm));
    set_scalars(atm, vatm);

    // The reference's nearest-texel table reads (atmosphere_kernels.cu:157-169, 375-395, 604-616) are not bounds-checked
    // and step past the end of a table for coordinates equal to 1.  Every table therefore sits between guard regions
    // filled with `guard_fill`: running twice with different fills tells which output texels depend on those reads.
    const size_t guard = 3 * (size_t)SCATTERING_TEXTURE_WIDTH * SCATTERING_TEXTURE_HEIGHT;
    const size_t sizes[9] = {nt, ni, ni, ns, ns, ns, ns, ns, ns};
    size_t arena_n = guard;
    for (size_t n : sizes) arena_n += n + guard;
    std::vector<float4> arena(arena_n, make_float4(guard_fill, guard_fill, guard_fill, guard_fill));
    float4* ptr[9];
    {
        size_t at = guard;
        for (int i = 0; i < 9; ++i) {
            ptr[i] = arena.data() + at;
            std::fill(ptr[i], ptr[i] + sizes[i], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            at += sizes[i] + guard;
        }
    }
    float4 *b_t = ptr[0], *b_i = ptr[2], *b_s = ptr[5], *b_sm = ptr[6], *b_dd = ptr[7];
    atm.transmittance_buffer = ptr[0];
    atm.delta_irradience_buffer = ptr[1];
    atm.irradiance_buffer = ptr[2];
    atm.delta_rayleigh_scattering_buffer = ptr[3];
    atm.delta_mie_scattering_buffer = ptr[4];
    atm.scattering_buffer = ptr[5];
    atm.optional_mie_single_scattering_buffer = ptr[6];
    atm.delta_scattering_density_buffer = ptr[7];
    atm.delta_multiple_scattering_buffer = ptr[8];

    static double identity[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};   // kDefaultLuminanceFromRadiance
    const int SZ = (max_z > 0 && max_z < SCATTERING_TEXTURE_DEPTH) ? max_z : SCATTERING_TEXTURE_DEPTH;
    const int runs = npasses > 0 ? npasses : 1;
    for (int pass = 0; pass < runs; ++pass) {
    double m9[9];
    std::memcpy(m9, npasses > 0 ? lfrms + 9 * pass : identity, sizeof(m9));
    mat3 lfrm;
    lfrm = lfrm.toMatrix(m9);
    const int BLEND = pass > 0 ? 1 : 0;                                           // init(): precompute(lambdas, luminance_from_radiance, i > 0, 4)
    if (npasses > 0) set_scalars(atm, passes + pass);                             // update_model(lambdas) of the pass

    launch(TRANSMITTANCE_TEXTURE_WIDTH, TRANSMITTANCE_TEXTURE_HEIGHT, 1, threads, [&]() { calculate_transmittance(atm); });
    launch(IRRADIANCE_TEXTURE_WIDTH, IRRADIANCE_TEXTURE_HEIGHT, 1, threads, [&]() { calculate_direct_irradiance(atm, BLEND); });
    {
        const float4 blend_vec = make_float4(0.0f, 0.0f, (float)BLEND, (float)BLEND);
        launch(SCATTERING_TEXTURE_WIDTH, SCATTERING_TEXTURE_HEIGHT, SCATTERING_TEXTURE_DEPTH, threads,
               [&]() { calculate_single_scattering(atm, blend_vec, lfrm); });
    }
    for (int order = 2; order <= num_scattering_orders; ++order) {
        const float4 density_blend = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        launch(SCATTERING_TEXTURE_WIDTH, SCATTERING_TEXTURE_HEIGHT, SZ, threads, [&]() { calculate_scattering_density(atm, density_blend, order); });
        // the host passes &float4(0,1,0,0) where the kernels declare `const int blend`: they read the bits of 0.0f
        const float4 blend_vec = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
        int blend_as_int;
        std::memcpy(&blend_as_int, &blend_vec, sizeof(int));
        launch(IRRADIANCE_TEXTURE_WIDTH, IRRADIANCE_TEXTURE_HEIGHT, 1, threads, [&]() { calculate_indirect_irradiance(atm, blend_as_int, lfrm, order); });
        launch(SCATTERING_TEXTURE_WIDTH, SCATTERING_TEXTURE_HEIGHT, SZ, threads, [&]() { calculate_multiple_scattering(atm, blend_as_int, lfrm, order); });
    }
    }   // passes
    if (npasses > 0) {
        // atmosphere::compute_transmittance (:1118-1175): the transmittance table once more, for the final wavelengths
        set_scalars(atm, vatm);
        launch(TRANSMITTANCE_TEXTURE_WIDTH, TRANSMITTANCE_TEXTURE_HEIGHT, 1, threads, [&]() { calculate_transmittance(atm); });
    }

    if (transmittance) std::memcpy(transmittance, b_t, nt * sizeof(float4));
    if (irradiance) std::memcpy(irradiance, b_i, ni * sizeof(float4));
    if (scattering) std::memcpy(scattering, b_s, ns * sizeof(float4));
    if (single_mie) std::memcpy(single_mie, b_sm, ns * sizeof(float4));
    if (delta_scattering_density) std::memcpy(delta_scattering_density, b_dd, ns * sizeof(float4));
    return 0;
}
