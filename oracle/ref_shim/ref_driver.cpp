// TEST INFRASTRUCTURE -- oracle/_ref/libvptref.so: the reference's OWN kernel source
// (/root/reference/source/render_kernel.cu and the headers it includes, plus source/bvh/octree.cpp), compiled
// unmodified by g++ where it lies, behind the same C entry point as the oracle's orc_render.  It exists to pin the
// oracle restatement (tests/test_oracle_vs_ref.py, tools/make_ref_golden.py): same scene in, same buffers out.
//
// This file contains no reference code.  It (1) includes the reference kernel as one translation unit over the
// stand-in CUDA headers of this directory, (2) converts the vpt_abi.h PODs to the reference's own classes field by
// field, (3) sets the octree root up the way the reference's host does before it builds the tree
// (source/bvh/bvh_builder.cpp:58-78), builds the tree with the reference's OCTree::create_tree (octree.cpp) and
// completes the fields only the device-side twin of that builder sets (see complete_children), and
// (4) calls volume_rt_kernel once per pixel with blockIdx set to that pixel.
//
// Two things the CUDA launch leaves undefined are given the oracle's definition here:
//   - the in-kernel blue-noise update (render_kernel.cu:2319-2325) races with the reads of other pixels of the same
//     launch; here every pixel of a launch reads the pre-launch table and the updates land after the launch;
//   - nothing else: values read from never-written buffers are whatever the caller put there.
#include "cuda_runtime.h"
thread_local uint3 blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;

#include "render_kernel.cu"      // the reference kernel (found through -I/root/reference/source)
#include "bvh/octree.h"

#undef rand
#include "../../include/vpt_abi.h"
#include <atomic>
#include <thread>

// the three special members gpu_vdb.h declares and the (OpenVDB-dependent) gpu_vdb.cpp would define
GPU_VDB::GPU_VDB() {}
GPU_VDB::~GPU_VDB() {}
GPU_VDB::GPU_VDB(const GPU_VDB& other) : vdb_info(other.vdb_info), xform(other.xform) {}

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

void free_tree(OCTNode* n, int depth) {
    if (!n || depth == 0 || n->num_volumes == 0) return;
    for (int i = 0; i < 8; ++i) {
        free_tree(n->children[i], depth - 1);
        delete n->children[i];
    }
}

// OCTree::create_tree is the reference's CPU builder; the builder it actually runs (the device-side twin,
// source/bvh/bvh_kernels.cu:204-246, which needs thrust and cannot be compiled here) fills four more fields per
// child: depth, has_children (num_volumes > 0), and the min of min_density / voxelsize over the node's volumes.
// volume_rt_kernel reads has_children of inner nodes (get_quadrant), so they are completed here.
void complete_children(OCTNode* n, int depth, const std::vector<GPU_VDB>& vdbs) {
    if (depth == 0 || n->num_volumes == 0) return;
    for (int i = 0; i < 8; ++i) {
        OCTNode* c = n->children[i];
        c->depth = depth;
        for (int k = 0; k < c->num_volumes; ++k) {
            const VDB_INFO& v = vdbs[c->vol_indices[k]].vdb_info;
            c->min_extinction = fminf(c->min_extinction, v.min_density);
            c->voxel_size = fminf(c->voxel_size, v.voxelsize);
        }
        if (c->num_volumes > 0) c->has_children = true;
        complete_children(c, depth - 1, vdbs);
    }
}

}  // namespace

extern "C" int ref_render(const vpt_camera* vcam, const vpt_light_list* vlights, const vpt_gpu_vdb* vvolumes,
                          int num_volumes, const vpt_sphere* vsphere, const vpt_atmosphere_parameters* vatm,
                          const vpt_kernel_params* vkp, unsigned int iter_count, int nthreads) {
    if (!vcam || !vlights || !vvolumes || !vsphere || !vatm || !vkp || num_volumes < 1 || num_volumes > 600) return -1;

    camera cam;
    cam.time1 = vcam->time1; cam.time0 = vcam->time0;
    cam.origin = cv(vcam->origin);
    cam.focus_dist = vcam->focus_dist;
    cam.lower_left_corner = cv(vcam->lower_left_corner);
    cam.horizontal = cv(vcam->horizontal);
    cam.vertical = cv(vcam->vertical);
    cam.u = cv(vcam->u); cam.v = cv(vcam->v); cam.w = cv(vcam->w);
    cam.lens_radius = vcam->lens_radius;
    cam.viz_dof = vcam->viz_dof != 0;

    std::vector<point_light> light_store(vlights->num_lights);
    for (unsigned i = 0; i < vlights->num_lights; ++i) {
        light_store[i].pos = cv(vlights->light_ptr[i].pos);
        light_store[i].dir = cv(vlights->light_ptr[i].dir);
        light_store[i].power = vlights->light_ptr[i].power;
        light_store[i].color = cv(vlights->light_ptr[i].color);
    }
    light_list lights(vlights->num_lights);
    lights.light_ptr = light_store.data();

    std::vector<GPU_VDB> vdbs(num_volumes);
    for (int i = 0; i < num_volumes; ++i) {
        const vpt_vdb_info& s = vvolumes[i].vdb_info;
        VDB_INFO& d = vdbs[i].vdb_info;
        d.voxelsize = s.voxelsize;
        d.dim = make_int3(s.dim.x, s.dim.y, s.dim.z);
        d.bmin = cv(s.bmin); d.bmax = cv(s.bmax);
        d.max_density = s.max_density; d.min_density = s.min_density;
        d.has_color = s.has_color != 0; d.has_emission = s.has_emission != 0; d.matte = s.matte != 0;
        d.density_texture = s.density_texture;
        d.emission_texture = s.emission_texture;
        d.color_texture = s.color_texture;
        mat4 m;
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m.m[c][r] = vvolumes[i].xform[c][r];
        vdbs[i].set_xform(m);
    }

    sphere ref_sphere;
    ref_sphere.center = cv(vsphere->center);
    ref_sphere.radius = vsphere->radius;
    ref_sphere.color = cv(vsphere->color);
    ref_sphere.roughness = vsphere->roughness;
    geometry_list geo_list;

    // octree root as the reference's host prepares it, then the reference's own recursive builder
    OCTree tree;
    OCTNode* root = new OCTNode;
    root->depth = 4;
    for (int i = 0; i < num_volumes; ++i) {
        const AABB b = vdbs[i].Bounds();
        root->bbox.pmax = fmaxf(root->bbox.pmax, b.pmax);
        root->bbox.pmin = fminf(root->bbox.pmin, b.pmin);
        root->vol_indices[i] = i;
        root->num_volumes++;
        root->max_extinction = fmaxf(root->max_extinction, vdbs[i].vdb_info.max_density);
        root->min_extinction = fminf(root->min_extinction, vdbs[i].vdb_info.min_density);
        root->has_children = true;
    }
    root->bbox.pmax += make_float3(1.0f);
    root->bbox.pmin -= make_float3(1.0f);
    tree.root_node = root;
    tree.create_tree(vdbs, root, 3);
    complete_children(root, 3, vdbs);

    AtmosphereParameters atm;
    std::memset(&atm, 0, sizeof(atm));
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
    atm.transmittance_texture = vatm->transmittance_texture;
    atm.scattering_texture = vatm->scattering_texture;
    atm.irradiance_texture = vatm->irradiance_texture;
    atm.single_mie_scattering_texture = vatm->single_mie_scattering_texture;

    Kernel_params kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.render = vkp->render != 0;
    kp.debug = vkp->debug != 0;
    kp.resolution = make_uint2(vkp->resolution.x, vkp->resolution.y);
    kp.exposure_scale = vkp->exposure_scale;
    kp.display_buffer = vkp->display_buffer;
    kp.raw_buffer = reinterpret_cast<float4*>(vkp->raw_buffer);
    kp.emission_texture = reinterpret_cast<float3*>(vkp->emission_texture);
    kp.emission_scale = vkp->emission_scale;
    kp.emission_pivot = vkp->emission_pivot;
    kp.density_color_texture = reinterpret_cast<float3*>(vkp->density_color_texture);
    kp.accum_buffer = reinterpret_cast<float3*>(vkp->accum_buffer);
    kp.depth_buffer = vkp->depth_buffer;
    kp.max_interactions = vkp->max_interactions;
    kp.ray_depth = vkp->ray_depth;
    kp.volume_depth = vkp->volume_depth;
    kp.min_extinction = vkp->min_extinction;
    kp.phase_g1 = vkp->phase_g1; kp.phase_g2 = vkp->phase_g2; kp.phase_f = vkp->phase_f;
    kp.albedo = cv(vkp->albedo);
    kp.extinction = cv(vkp->extinction);
    kp.transmittance = cv(vkp->transmittance);
    kp.tr_depth = vkp->tr_depth;
    kp.density_mult = vkp->density_mult;
    kp.environment_type = vkp->environment_type;
    kp.azimuth = vkp->azimuth;
    kp.elevation = vkp->elevation;
    kp.sun_color = cv(vkp->sun_color);
    kp.sky_color = cv(vkp->sky_color);
    kp.sun_mult = vkp->sun_mult;
    kp.sky_mult = vkp->sky_mult;
    kp.energy_inject = vkp->energy_inject;
    kp.env_tex = vkp->env_tex;
    kp.env_sample_tex_res = vkp->env_sample_tex_res;
    kp.sky_tex = vkp->sky_tex;
    kp.env_func_tex = vkp->env_func_tex;
    kp.env_cdf_tex = vkp->env_cdf_tex;
    kp.env_marginal_func_tex = vkp->env_marginal_func_tex;
    kp.env_marginal_cdf_tex = vkp->env_marginal_cdf_tex;
    kp.env_marginal_int = vkp->env_marginal_int;
    kp.debug_buffer = reinterpret_cast<float3*>(vkp->debug_buffer);
    kp.cost_buffer = reinterpret_cast<float3*>(vkp->cost_buffer);
    kp.integrator = vkp->integrator;

    const int W = (int)kp.resolution.x, H = (int)kp.resolution.y;
    const int BN = 256 * 256;
    float3* bn_table = reinterpret_cast<float3*>(vkp->blue_noise_buffer);
    std::vector<float3> bn_next(BN);
    const int threads = nthreads > 1 ? nthreads : 1;

    for (unsigned int k = 0; k < iter_count; ++k) {
        kp.iteration = vkp->iteration + k;
        for (int i = 0; i < BN; ++i) bn_next[i] = bn_table[i];
        std::atomic<int> cursor{0};
        auto worker = [&]() {
            // a private table of which the kernel only ever touches two entries per pixel: the one it reads
            // ((y%256)*256 + x%256) and, for the first 65536 pixels, the one it advances (its own index)
            std::vector<float3> bn_private(BN);
            Kernel_params kpt = kp;
            kpt.blue_noise_buffer = bn_private.data();
            blockDim = dim3(1, 1, 1);
            gridDim = dim3((unsigned)W, (unsigned)H, 1);
            threadIdx = make_uint3(0, 0, 0);
            for (;;) {
                const int first = cursor.fetch_add(64);
                if (first >= W * H) break;
                const int last = first + 64 < W * H ? first + 64 : W * H;
                for (int p = first; p < last; ++p) {
                    const int x = p % W, y = p / W;
                    const int src = (y % 256) * 256 + (x % 256);
                    bn_private[src] = bn_table[src];
                    if (p < BN) bn_private[p] = bn_table[p];
                    blockIdx = make_uint3((unsigned)x, (unsigned)y, 0);
                    volume_rt_kernel(cam, lights, vdbs.data(), ref_sphere, geo_list, nullptr, root, atm, kpt);
                    if (p < BN) bn_next[p] = bn_private[p];
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        for (int i = 0; i < BN; ++i) bn_table[i] = bn_next[i];
    }

    free_tree(root, 3);
    delete root;
    return 0;
}

// cuRAND stream through the stand-in (cross-checked against the oracle's own stream in the tests)
extern "C" void ref_curand_uniform_stream(unsigned long long seed, unsigned long long offset, int n, float* out) {
    curandStatePhilox4_32_10_t s;
    curand_init(seed, 0, offset, &s);
    for (int i = 0; i < n; ++i) out[i] = curand_uniform(&s);
}

// ---- host-side helpers of the reference, for pinning the product's restatements (vpt_camera_update,
// vpt_gpu_vdb_bounds, vpt_instance_xform) ----------------------------------------------------------------------------
extern "C" void ref_camera_update(vpt_camera* out, const float lookfrom[3], const float lookat[3], const float vup[3],
                                  float vfov, float aspect, float aperture) {
    camera c;                                                  // the reference's default constructor, then its method
    c.update_camera(make_float3(lookfrom[0], lookfrom[1], lookfrom[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                    make_float3(vup[0], vup[1], vup[2]), vfov, aspect, aperture);
    auto st = [](vpt_float3& d, float3 v) { d.x = v.x; d.y = v.y; d.z = v.z; };
    out->time1 = c.time1; out->time0 = c.time0;
    st(out->origin, c.origin);
    out->focus_dist = c.focus_dist;
    st(out->lower_left_corner, c.lower_left_corner);
    st(out->horizontal, c.horizontal);
    st(out->vertical, c.vertical);
    st(out->u, c.u); st(out->v, c.v); st(out->w, c.w);
    out->lens_radius = c.lens_radius;
}

extern "C" void ref_gpu_vdb_bounds(const vpt_gpu_vdb* v, float pmin[3], float pmax[3]) {
    GPU_VDB g;
    g.vdb_info.bmin = cv(v->vdb_info.bmin);
    g.vdb_info.bmax = cv(v->vdb_info.bmax);
    mat4 m;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m.m[c][r] = v->xform[c][r];
    g.set_xform(m);
    const AABB b = g.Bounds();
    pmin[0] = b.pmin.x; pmin[1] = b.pmin.y; pmin[2] = b.pmin.z;
    pmax[0] = b.pmax.x; pmax[1] = b.pmax.y; pmax[2] = b.pmax.z;
}

// the calls the reference's .ins loader makes on a file's matrix for one instance (source/main.cpp:1060-1095), on the
// reference's own mat4 (matrix_math.h)
extern "C" void ref_instance_xform(const float base[4][4], const double position[3], const double rotation[4], double scale,
                                   float out[4][4]) {
    mat4 xform;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) xform.m[c][r] = base[c][r];
    xform.translate(-xform.extract_translate());
    xform.scale(make_float3(scale));
    mat4 rotation_matrix = quaternion_to_mat4(rotation[0], rotation[1], rotation[2], rotation[3]);
    xform = rotation_matrix * xform;
    xform.translate(make_float3(position[0], position[1], position[2]));
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out[c][r] = xform.m[c][r];
}
