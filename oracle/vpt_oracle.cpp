// vpt_oracle.cpp -- TEST INFRASTRUCTURE: CPU restatement of the reference hot path
// `volume_rt_kernel` (reference source/render_kernel.cu:2216) and every function it
// calls.  One C++ thread = one CUDA thread of the reference; control flow, operand
// order and the quirks listed in SURVEY.md 8a ("Q-list") are kept on purpose.
//
// PARITY PINNED on the reference's own kernel source compiled for the CPU (oracle/_ref, see vpt_oracle.h and
// tests/test_oracle_vs_ref.py).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build, load or call this file.
//
// Third-party arithmetic that is not under /root/reference and is restated here:
//   * cuRAND Philox4_32_10 (CUDA 12.1): D. E. Shaw Research Random123 philox4x32-10,
//     key=(seed_lo,seed_hi), counter += offset/4, 4 outputs per block,
//     curand_uniform(x) = x*2^-32 + 2^-33   (call sites render_kernel.cu:2235,
//     gpu_vdb/camera.h:45-46,134, light.h:51-52)
//   * CUDA texture unit: addressing per the CUDA C Programming Guide appendix
//     "Texture Fetching" (xB = u*N - 0.5, clamp / wrap), weights kept in full fp32
//     instead of the hardware's 1.8 fixed point (documented deviation)
//   * __logf/__sinf/__cosf of --use_fast_math: replaced by orc_math.h's deterministic
//     routines on the decision path, libm elsewhere.
#include "vpt_oracle.h"
#include "orc_math.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

static const float EPS = 0.001f;                              // render_kernel.cu:83
static inline f3 BLACK() { return mk3(0.0f); }
static inline f3 WHITE() { return mk3(1.0f); }
static inline f3 to_f3(vpt_float3 v) { return mk3(v.x, v.y, v.z); }
static inline vpt_float3 to_v3(f3 v) { vpt_float3 r = {v.x, v.y, v.z}; return r; }

// ------------------------------------------------------------------------------------
// Textures (CUDA sampler semantics)
// ------------------------------------------------------------------------------------
struct Tex {
    vpt_texture_desc d;
    const float* data;
};

static inline int addr(int i, int n, int mode, int normalized) {
    // wrap is only honoured for normalised coordinates (CUDA Programming Guide:
    // "cudaAddressModeWrap ... only supported for normalized texture coordinates")
    if (mode == VPT_ADDR_WRAP && normalized) {
        int r = i % n;
        return r < 0 ? r + n : r;
    }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}

static inline f4 texel(const Tex* t, int x, int y, int z) {
    size_t idx = ((size_t)z * t->d.height + y) * t->d.width + x;
    if (t->d.channels == 1) return mk4(t->data[idx], 0.0f, 0.0f, 0.0f);
    const float* p = t->data + idx * 4;
    return mk4(p[0], p[1], p[2], p[3]);
}

static inline f4 lerp4(f4 a, f4 b, float t) { return a + (b - a) * t; }

struct AxisTap { int i0, i1; float a; };

static inline AxisTap axis_tap(float u, int n, int mode, int normalized, int linear) {
    AxisTap r;
    float x = normalized ? u * (float)n : u;
    if (linear) {
        float xb = x - 0.5f;
        float fl = std::floor(xb);
        r.a = xb - fl;
        int i = (int)fl;
        r.i0 = addr(i, n, mode, normalized);
        r.i1 = addr(i + 1, n, mode, normalized);
    } else {
        int i = (int)std::floor(x);
        r.a = 0.0f;
        r.i0 = r.i1 = addr(i, n, mode, normalized);
    }
    return r;
}

// DIAGNOSTIC switch (orc_set_volume_tex_weights; never on in the parity tests): the CUDA texture unit holds the linear filter's weights in 9-bit
// fixed point with 8 fractional bits (CUDA C Programming Guide, "Linear Filtering"); the reference's tex3D calls on the volume grids
// (render_kernel.cu:999-1014) run on that hardware, while the parity contract -- the reference compiled for the CPU, this oracle, the HIP path -- is full
// binary32 weights.  Model of the hardware, for the three volume-grid look-ups only: weight = floor(a * 256 + 0.5) / 256, the same nested lerps.
static int g_volume_weight_bits = 32;
static inline float quant8(float a) { return std::floor(a * 256.0f + 0.5f) * 0.00390625f; }

static f4 tex_sample(const Tex* t, float u, float v, float w, bool volume_grid = false) {
    const int lin = t->d.filter_mode == VPT_FILTER_LINEAR;
    const int nrm = t->d.normalized_coords;
    AxisTap ax = axis_tap(u, t->d.width, t->d.address_mode[0], nrm, lin);
    AxisTap ay = {0, 0, 0.0f}, az = {0, 0, 0.0f};
    if (t->d.height > 1 || t->d.depth > 1) ay = axis_tap(v, t->d.height, t->d.address_mode[1], nrm, lin);
    if (t->d.depth > 1) az = axis_tap(w, t->d.depth, t->d.address_mode[2], nrm, lin);
    if (volume_grid && g_volume_weight_bits == 8) { ax.a = quant8(ax.a); ay.a = quant8(ay.a); az.a = quant8(az.a); }
    if (!lin) return texel(t, ax.i0, ay.i0, az.i0);
    // nested lerp: x, then y, then z; lerp(a,b,t) = a + t*(b-a)
    f4 c00 = lerp4(texel(t, ax.i0, ay.i0, az.i0), texel(t, ax.i1, ay.i0, az.i0), ax.a);
    if (t->d.height == 1 && t->d.depth == 1) return c00;
    f4 c10 = lerp4(texel(t, ax.i0, ay.i1, az.i0), texel(t, ax.i1, ay.i1, az.i0), ax.a);
    f4 c0 = lerp4(c00, c10, ay.a);
    if (t->d.depth == 1) return c0;
    f4 c01 = lerp4(texel(t, ax.i0, ay.i0, az.i1), texel(t, ax.i1, ay.i0, az.i1), ax.a);
    f4 c11 = lerp4(texel(t, ax.i0, ay.i1, az.i1), texel(t, ax.i1, ay.i1, az.i1), ax.a);
    f4 c1 = lerp4(c01, c11, ay.a);
    return lerp4(c0, c1, az.a);
}

static inline const Tex* as_tex(vpt_texture_t h) { return reinterpret_cast<const Tex*>((uintptr_t)h); }
static inline float tex1(vpt_texture_t h, float u) { return tex_sample(as_tex(h), u, 0.0f, 0.0f).x; }
static inline float tex2(vpt_texture_t h, float u, float v) { return tex_sample(as_tex(h), u, v, 0.0f).x; }
static inline f4 tex2_4(vpt_texture_t h, float u, float v) { return tex_sample(as_tex(h), u, v, 0.0f); }
static inline f4 tex3_4(vpt_texture_t h, float u, float v, float w) { return tex_sample(as_tex(h), u, v, w); }

// ------------------------------------------------------------------------------------
// Philox4x32-10 with cuRAND stream semantics
// ------------------------------------------------------------------------------------
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

static inline void philox10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

struct Rng {                       // curandStatePhilox4_32_10_t (camera.h:44)
    uint32_t ctr[4];
    uint32_t key[2];
    uint32_t out[4];
    uint32_t idx;                  // STATE
    uint64_t draws;                // bookkeeping only
};

static inline void rng_incr(Rng* s, uint64_t n) {       // Philox_State_Incr: 128-bit counter += n
    const uint64_t lo = ((uint64_t)s->ctr[1] << 32) | s->ctr[0];
    const uint64_t nlo = lo + n;
    s->ctr[0] = (uint32_t)nlo;
    s->ctr[1] = (uint32_t)(nlo >> 32);
    if (nlo < lo) {
        if (++s->ctr[2] == 0) ++s->ctr[3];
    }
}

// curand_init(seed, subsequence=0, offset, &state)  (render_kernel.cu:2235)
static inline void rng_init(Rng* s, uint64_t seed, uint64_t offset) {
    s->ctr[0] = s->ctr[1] = s->ctr[2] = s->ctr[3] = 0;
    s->key[0] = (uint32_t)seed;
    s->key[1] = (uint32_t)(seed >> 32);
    s->idx = (uint32_t)(offset & 3);
    rng_incr(s, offset / 4);
    philox10(s->ctr, s->key, s->out);
    s->draws = 0;
}

static inline uint32_t rng_next(Rng* s) {
    uint32_t r = s->out[s->idx++];
    if (s->idx == 4) {
        rng_incr(s, 1);
        philox10(s->ctr, s->key, s->out);
        s->idx = 0;
    }
    s->draws++;
    return r;
}

// curand_uniform: (0, 1]
static inline float rnd(Rng* s) {
    return (float)rng_next(s) * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

// ------------------------------------------------------------------------------------
// AABB / sphere / octree   (bvh/AABB.h, geometry/geometry.h, bvh/bvh_kernels.cu)
// ------------------------------------------------------------------------------------
struct Box { f3 pmin, pmax; };

static inline bool contains(const Box& b, f3 p) {           // AABB.h:141-146
    return (p.x >= b.pmin.x && p.x <= b.pmax.x && p.y >= b.pmin.y && p.y <= b.pmax.y &&
            p.z >= b.pmin.z && p.z <= b.pmax.z);
}
static inline bool overlaps(const Box& a, const Box& b) {   // AABB.h:134-139
    bool x = (a.pmax.x >= b.pmin.x) && (a.pmin.x <= b.pmax.x);
    bool y = (a.pmax.y >= b.pmin.y) && (a.pmin.y <= b.pmax.y);
    bool z = (a.pmax.z >= b.pmin.z) && (a.pmin.z <= b.pmax.z);
    return (x && y && z);
}
// AABB::Intersect, AABB.h:182-205
static inline bool box_intersect(const Box& b, f3 o, f3 d, float& tmin, float& tmax) {
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float t1 = (b.pmin.x - o.x) * inv.x;
    float t2 = (b.pmax.x - o.x) * inv.x;
    float t3 = (b.pmin.y - o.y) * inv.y;
    float t4 = (b.pmax.y - o.y) * inv.y;
    float t5 = (b.pmin.z - o.z) * inv.z;
    float t6 = (b.pmax.z - o.z) * inv.z;
    tmin = fmaxf_(fmaxf_(fminf_(t1, t2), fminf_(t3, t4)), fminf_(t5, t6));
    tmax = fminf_(fminf_(fmaxf_(t1, t2), fmaxf_(t3, t4)), fmaxf_(t5, t6));
    if (tmax <= 0.0f) return false;
    if (tmin > tmax) return false;
    if (tmin < 0) {
        tmin = tmax;
        if (tmin < 0) return false;
    }
    return true;
}

// find_discr, geometry.h:46-70
static inline bool find_discr(float a, float b, float c, float& x1, float& x2) {
    if (b == 0) {
        if (a == 0) return false;
        x1 = 0;
        x2 = sqrtf(-c / a);
        return true;
    }
    float discr = b * b - 4 * a * c;
    if (discr < 0) return false;
    float q = (b < 0.f) ? -0.5f * (b - sqrtf(discr)) : -0.5f * (b + sqrtf(discr));
    x1 = q / a;
    x2 = c / q;
    return true;
}
// sphere::intersect, geometry.h:114-137
static inline int sphere_intersect(const vpt_sphere& s, f3 ray_pos, f3 ray_dir, float& t_min, float& t_max) {
    f3 orig = ray_pos - to_f3(s.center);
    float A = ray_dir.x * ray_dir.x + ray_dir.y * ray_dir.y + ray_dir.z * ray_dir.z;
    float B = 2 * (ray_dir.x * orig.x + ray_dir.y * orig.y + ray_dir.z * orig.z);
    float C = orig.x * orig.x + orig.y * orig.y + orig.z * orig.z - s.radius * s.radius;
    if (!find_discr(A, B, C, t_min, t_max)) return 0;
    if (t_min > t_max) {
        float tempt = t_max;
        t_max = t_min;
        t_min = tempt;
    }
    if (t_min < 0) {
        t_min = t_max;
        if (t_min < 0) return 0;
    }
    return 1;
}

struct Node {                                               // OCTNode, AABB.h:217-234
    int num_volumes = 0;
    std::vector<int> vol_indices;
    float max_extinction = 0.0f;
    float min_extinction = kInf;
    float voxel_size = kInf;
    int depth = -1;
    bool has_children = false;
    Node* children[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Box bbox = {mk3(kInf), mk3(-kInf)};
};

struct Volume {
    vpt_gpu_vdb vdb;
    mat4 w2i;           // xform.transpose().inverse()  (render_kernel.cu:987)
    Box bounds;         // GPU_VDB::Bounds()
};

static mat4 load_xform(const vpt_gpu_vdb& v) {
    mat4 m;
    std::memcpy(m.m, v.xform, sizeof(m.m));
    return m;
}

// GPU_VDB::Bounds, gpu_vdb.h:131-146
static Box vdb_bounds(const vpt_gpu_vdb& v) {
    f3 bmin = to_f3(v.vdb_info.bmin), bmax = to_f3(v.vdb_info.bmax);
    // (bmax + bmin) * 0.5 : float3 * double literal -> float (helper_math float3*float)
    f3 center = (bmax + bmin) * 0.5f;
    f3 extent = (bmax - bmin) * 0.5f;
    mat4 x = load_xform(v);
    f3 nc = mat4_transform_point(mat4_transpose(x), center);
    f3 ne = mat4_transform_vector(mat4_transpose(mat4_abs(x)), extent);
    Box b = {nc - ne, nc + ne};
    return b;
}

// divide_bbox, bvh_kernels.cu:150-202
static Box divide_bbox(int idx, f3 pmin, f3 pmax) {
    f3 mn = mk3(0.0f), mx = mk3(0.0f);
    float hx = (float)((pmin.x + pmax.x) * 0.5);
    float hy = (float)((pmin.y + pmax.y) * 0.5);
    float hz = (float)((pmin.z + pmax.z) * 0.5);
    switch (idx) {
        case 0: mn = mk3(pmin.x, hy, pmin.z); mx = mk3(hx, pmax.y, hz); break;
        case 1: mn = mk3(hx, hy, pmin.z);     mx = mk3(pmax.x, pmax.y, hz); break;
        case 2: mn = pmin;                    mx = mk3(hx, hy, hz); break;
        case 3: mn = mk3(hx, pmin.y, pmin.z); mx = mk3(pmax.x, hy, hz); break;
        case 4: mn = mk3(pmin.x, hy, hz);     mx = mk3(hx, pmax.y, pmax.z); break;
        case 5: mn = mk3(hx, hy, hz);         mx = pmax; break;
        case 6: mn = mk3(pmin.x, pmin.y, hz); mx = mk3(hx, hy, pmax.z); break;
        case 7: mn = mk3(hx, pmin.y, hz);     mx = mk3(pmax.x, hy, pmax.z); break;
    }
    Box b = {mn, mx};
    return b;
}

struct Scene {
    std::vector<Volume> vols;
    std::vector<std::unique_ptr<Node>> pool;
    Node* root = nullptr;

    Node* alloc() { pool.emplace_back(new Node()); return pool.back().get(); }

    // build_octree_recursive, bvh_kernels.cu:204-246
    void build_rec(Node* n, int depth) {
        if (depth > 0) {
            if (n->num_volumes > 0) {
                for (int i = 0; i < 8; ++i) {
                    Node* c = alloc();
                    n->children[i] = c;
                    c->depth = depth;
                    c->bbox = divide_bbox(i, n->bbox.pmin, n->bbox.pmax);
                    for (size_t y = 0; y < vols.size(); ++y) {
                        if (overlaps(c->bbox, vols[y].bounds)) {
                            c->num_volumes++;
                            c->vol_indices.push_back((int)y);
                            c->max_extinction = fmaxf_(c->max_extinction, vols[y].vdb.vdb_info.max_density);
                            c->min_extinction = fminf_(c->min_extinction, vols[y].vdb.vdb_info.min_density);
                            c->voxel_size = fminf_(c->voxel_size, vols[y].vdb.vdb_info.voxelsize);
                        }
                    }
                    if (c->num_volumes > 0) c->has_children = true;
                    build_rec(c, depth - 1);
                }
            }
        }
    }

    // BVH_Builder::build_bvh (octree part), bvh_builder.cpp:61-96
    void build(const vpt_gpu_vdb* v, int n) {
        vols.resize(n);
        for (int i = 0; i < n; ++i) {
            vols[i].vdb = v[i];
            vols[i].w2i = mat4_inverse(mat4_transpose(load_xform(v[i])));
            vols[i].bounds = vdb_bounds(v[i]);
        }
        root = alloc();
        root->depth = 4;
        for (int i = 0; i < n; ++i) {
            root->bbox.pmax = fmax3(root->bbox.pmax, vols[i].bounds.pmax);
            root->bbox.pmin = fmin3(root->bbox.pmin, vols[i].bounds.pmin);
            root->vol_indices.push_back(i);
            root->num_volumes++;
            root->max_extinction = fmaxf_(root->max_extinction, v[i].vdb_info.max_density);
            root->min_extinction = fminf_(root->min_extinction, v[i].vdb_info.min_density);
            root->has_children = true;
        }
        root->bbox.pmax += mk3(1.0f);
        root->bbox.pmin -= mk3(1.0f);
        build_rec(root, root->depth - 1);
    }
};

// ------------------------------------------------------------------------------------
// Per-thread context
// ------------------------------------------------------------------------------------
struct Ctx {
    const Scene* sc;
    const vpt_kernel_params* kp;
    const vpt_atmosphere_parameters* atm;
    const vpt_sphere* sphere;
    const vpt_light_list* lights;
    orc_stats st;
};

// ------------------------------------------------------------------------------------
// Camera (gpu_vdb/camera.h)
// ------------------------------------------------------------------------------------
// vanDerCorput, camera.h:49-62
static float van_der_corput(Rng* s, int base) {
    int n = (int)(rnd(s) * 100);
    float rand_int = 0, denom = 1, invBase = 1.f / base;
    while (n) {
        denom *= base;
        rand_int += (n % base) / denom;
        n = (int)(n * invBase);
    }
    return rand_int;
}
// random_in_unit_disk, camera.h:65-75
static f3 random_in_unit_disk(Rng* s) {
    f3 p;
    do {
        float a = van_der_corput(s, 2);
        float b = van_der_corput(s, 3);
        p = 2.0f * mk3(a, b, 0) - mk3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0);
    return p;
}
// camera::get_ray, camera.h:131-136
static void get_ray(const vpt_camera& c, float s, float t, Rng* rng, f3& A, f3& B) {
    f3 rd = c.lens_radius * random_in_unit_disk(rng);
    f3 offset = to_f3(c.u) * rd.x + to_f3(c.v) * rd.y;
    float time = c.time0 + rnd(rng) * (c.time1 - c.time0);
    (void)time;
    A = to_f3(c.origin) + offset;
    B = to_f3(c.lower_left_corner) + s * to_f3(c.horizontal) + t * to_f3(c.vertical) - to_f3(c.origin) - offset;
}

// ------------------------------------------------------------------------------------
// Small helpers of render_kernel.cu
// ------------------------------------------------------------------------------------
// coordinate_system :92-102
static inline void coordinate_system(f3 v1, f3& v2, f3& v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y)) v2 = mk3(-v1.z, 0.0f, v1.x);
    else v2 = mk3(0.0f, v1.z, -v1.y);
    v2 = normalize(v2);
    v3 = normalize(cross(v1, v2));
}
// spherical_direction :104-115
static inline f3 spherical_direction(float sinTheta, float cosTheta, float phi, f3 x, f3 y, f3 z) {
    return x * sinTheta * orc_cosf(phi) + y * sinTheta * orc_sinf(phi) + z * cosTheta;
}
// degree_to_radians :117-123
static inline float degree_to_radians(float degree) { return degree * kPi / 180.0f; }
// degree_to_cartesian :126-142
static inline f3 degree_to_cartesian(float azimuth, float elevation) {
    float az = clampf(azimuth, .0f, 360.0f);
    float el = clampf(elevation, -90.0f, 90.0f);
    az = degree_to_radians(az);
    el = degree_to_radians(90.0f - el);
    float x = orc_sinf(el) * orc_cosf(az);
    float y = orc_cosf(el);
    float z = orc_sinf(el) * orc_sinf(az);
    return normalize(mk3(x, y, z));
}
// henyey_greenstein, light.h:55-64   (note the pi/4 normalisation, Q-list 4)
static inline float henyey_greenstein(float cos_theta, float g) {
    float denominator = 1 + g * g - 2 * g * cos_theta;
    return kPi4 * (1 - g * g) / (denominator * sqrtf(denominator));
}
// power_heuristic, light.h:65-69
static inline float power_heuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
static inline float isotropic() { return 1.0f / (4.0f * kPi); }   // :271-275

// sample_hg :306-325 (mutates wo, advances the caller's rng)
static float sample_hg(f3& wo, Rng& rng, float g) {
    float cos_theta;
    if (std::fabs(g) < EPS) cos_theta = 1 - 2 * rnd(&rng);
    else {
        float sqr_term = (1 - g * g) / (1 - g + 2 * g * rnd(&rng));
        cos_theta = (1 + g * g - sqr_term * sqr_term) / (2 * g);
    }
    float sin_theta = sqrtf(fmaxf_(.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * (double)kPi) * rnd(&rng);
    f3 v1, v2;
    coordinate_system(wo * -1.0f, v1, v2);
    wo = spherical_direction(sin_theta, cos_theta, phi, v1, v2, wo);
    return henyey_greenstein(-cos_theta, g);
}

// sample_spherical :292-303 (rng BY VALUE: caller's stream does not advance, Q-list 2)
static float sample_spherical(Rng rng, f3& wi) {
    float phi = (float)(2.0f * kPi) * rnd(&rng);
    float cos_theta = 1.0f - 2.0f * rnd(&rng);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    wi = mk3(orc_cosf(phi) * sin_theta, orc_sinf(phi) * sin_theta, cos_theta);
    return isotropic();
}

// ------------------------------------------------------------------------------------
// Bruneton atmosphere look-ups (render_kernel.cu:369-895).  Value-only code: nothing
// downstream branches on it, so libm is used for the transcendental functions.
// Literals without an f suffix are double in the reference and are kept double here.
// ------------------------------------------------------------------------------------
typedef vpt_atmosphere_parameters Atm;

static inline float ClampCosine(float mu) { return clampf(mu, -1.0f, 1.0f); }
static inline float ClampDistance(float d) { return fmaxf_(d, 0.0f); }
static inline float ClampRadius(const Atm& a, float r) { return clampf(r, a.bottom_radius, a.top_radius); }
static inline float SafeSqrt(float a) { return sqrtf(fmaxf_(a, 0.0f)); }

static inline float DistanceToTopAtmosphereBoundary(const Atm& a, float r, float mu) {       // :389
    float discriminant = (float)(r * r * (mu * mu - 1.0) + a.top_radius * a.top_radius);
    return ClampDistance(-r * mu + SafeSqrt(discriminant));
}
static inline bool RayIntersectsGround(const Atm& a, float r, float mu) {                      // :401
    return mu < 0.0 && r * r * (mu * mu - 1.0) + a.bottom_radius * a.bottom_radius >= 0.0;
}
static inline float GetTextureCoordFromUnitRange(float x, int texture_size) {                  // :419
    return (float)(0.5 / (float)texture_size + x * (1.0 - 1.0 / (float)texture_size));
}
static inline f2 GetTransmittanceTextureUvFromRMu(const Atm& a, float r, float mu) {           // :429
    float H = sqrtf(a.top_radius * a.top_radius - a.bottom_radius * a.bottom_radius);
    float rho = SafeSqrt(r * r - a.bottom_radius * a.bottom_radius);
    float d = DistanceToTopAtmosphereBoundary(a, r, mu);
    float d_min = a.top_radius - r;
    float d_max = rho + H;
    float x_mu = (d - d_min) / (d_max - d_min);
    float x_r = rho / H;
    f2 uv = {GetTextureCoordFromUnitRange(x_mu, VPT_TRANSMITTANCE_W), GetTextureCoordFromUnitRange(x_r, VPT_TRANSMITTANCE_H)};
    return uv;
}
static inline f3 GetTransmittanceToTopAtmosphereBoundary(const Atm& a, float r, float mu) {    // :464
    f2 uv = GetTransmittanceTextureUvFromRMu(a, r, mu);
    return xyz(tex2_4(a.transmittance_texture, uv.x, uv.y));
}
static inline f3 GetTransmittance(const Atm& a, float r, float mu, float d, bool hits_ground) { // :472
    float r_d = ClampRadius(a, (float)std::sqrt(d * d + 2.0 * r * mu * d + r * r));
    float mu_d = ClampCosine((r * mu + d) / r_d);
    if (hits_ground) {
        return fmin3(GetTransmittanceToTopAtmosphereBoundary(a, r_d, -mu_d) /
                         GetTransmittanceToTopAtmosphereBoundary(a, r, -mu), mk3(1.0f));
    } else {
        return fmin3(GetTransmittanceToTopAtmosphereBoundary(a, r, mu) /
                         GetTransmittanceToTopAtmosphereBoundary(a, r_d, mu_d), mk3(1.0f));
    }
}
static inline f3 GetTransmittanceToSun(const Atm& a, float r, float mu_s) {                     // :486
    float sin_theta_h = a.bottom_radius / r;
    float cos_theta_h = -sqrtf((float)std::max(1.0 - sin_theta_h * sin_theta_h, 0.0));
    return GetTransmittanceToTopAtmosphereBoundary(a, r, mu_s) *
           smoothstep(-sin_theta_h * a.sun_angular_radius, sin_theta_h * a.sun_angular_radius, mu_s - cos_theta_h);
}
static inline float RayleighPhaseFunction(float nu) {                                          // :508
    float k = (float)(3.0 / (16.0 * kPi));
    return (float)(k * (1.0 + nu * nu));
}
static inline float MiePhaseFunction(float g, float nu) {                                      // :514
    float k = (float)(3.0 / (8.0 * kPi) * (1.0 - g * g) / (2.0 + g * g));
    return (float)(k * (1.0 + nu * nu) / std::pow(1.0 + g * g - 2.0 * g * nu, 1.5));
}
// GetScatteringTextureUvwzFromRMuMuSNu :520-569
static inline f4 ScatteringUvwz(const Atm& a, float r, float mu, float mu_s, float nu, bool hits_ground) {
    float H = sqrtf(a.top_radius * a.top_radius - a.bottom_radius * a.bottom_radius);
    float rho = SafeSqrt(r * r - a.bottom_radius * a.bottom_radius);
    float u_r = GetTextureCoordFromUnitRange(rho / H, VPT_SCATTERING_R);
    float r_mu = r * mu;
    float discriminant = r_mu * r_mu - r * r + a.bottom_radius * a.bottom_radius;
    float u_mu;
    if (hits_ground) {
        float d = -r_mu - SafeSqrt(discriminant);
        float d_min = r - a.bottom_radius;
        float d_max = rho;
        u_mu = (float)(0.5 - 0.5 * GetTextureCoordFromUnitRange(
                                       d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min), VPT_SCATTERING_MU / 2));
    } else {
        float d = -r_mu + SafeSqrt(discriminant + H * H);
        float d_min = a.top_radius - r;
        float d_max = rho + H;
        u_mu = (float)(0.5 + 0.5 * GetTextureCoordFromUnitRange((d - d_min) / (d_max - d_min), VPT_SCATTERING_MU / 2));
    }
    float d = DistanceToTopAtmosphereBoundary(a, a.bottom_radius, mu_s);
    float d_min = a.top_radius - a.bottom_radius;
    float d_max = H;
    float aa = (d - d_min) / (d_max - d_min);
    float A = (float)(-2.0 * a.mu_s_min * a.bottom_radius / (d_max - d_min));
    float u_mu_s = GetTextureCoordFromUnitRange((float)(std::max(1.0 - aa / A, 0.0) / (1.0 + aa)), VPT_SCATTERING_MU_S);
    float u_nu = (float)((nu + 1.0) / 2.0);
    return mk4(u_nu, u_mu_s, u_mu, u_r);
}
static inline f2 GetIrradianceTextureUvFromRMuS(const Atm& a, float r, float mu_s) {           // :633
    float x_r = (r - a.bottom_radius) / (a.top_radius - a.bottom_radius);
    float x_mu_s = (float)(mu_s * 0.5 + 0.5);
    f2 uv = {GetTextureCoordFromUnitRange(x_mu_s, VPT_IRRADIANCE_W), GetTextureCoordFromUnitRange(x_r, VPT_IRRADIANCE_H)};
    return uv;
}
static inline f3 GetIrradiance(const Atm& a, float r, float mu_s) {                            // :649
    f2 uv = GetIrradianceTextureUvFromRMuS(a, r, mu_s);
    return xyz(tex2_4(a.irradiance_texture, uv.x, uv.y));
}
// GetCombinedScattering :672-692 (COMBINED_SCATTERING_TEXTURES off, :72)
static inline f3 GetCombinedScattering(const Atm& a, float r, float mu, float mu_s, float nu, bool hits_ground,
                                       f3& single_mie_scattering) {
    f4 uvwz = ScatteringUvwz(a, r, mu, mu_s, nu, hits_ground);
    float tex_coord_x = uvwz.x * (float)(VPT_SCATTERING_NU - 1);
    float tex_x = std::floor(tex_coord_x);
    float lerp = tex_coord_x - tex_x;
    f3 uvw0 = mk3((tex_x + uvwz.y) / (float)VPT_SCATTERING_NU, uvwz.z, uvwz.w);
    f3 uvw1 = mk3((float)((tex_x + 1.0 + uvwz.y) / (float)VPT_SCATTERING_NU), uvwz.z, uvwz.w);
    float l0 = (float)(1.0 - lerp);
    f3 scattering = xyz(tex3_4(a.scattering_texture, uvw0.x, uvw0.y, uvw0.z) * l0 +
                        tex3_4(a.scattering_texture, uvw1.x, uvw1.y, uvw1.z) * lerp);
    single_mie_scattering = xyz(tex3_4(a.single_mie_scattering_texture, uvw0.x, uvw0.y, uvw0.z) * l0 +
                                tex3_4(a.single_mie_scattering_texture, uvw1.x, uvw1.y, uvw1.z) * lerp);
    return scattering;
}
// GetSkyRadiance :694-747
static f3 GetSkyRadiance(const Atm& a, f3 camera, f3 view_ray, float shadow_length, f3 sun_direction, f3& transmittance) {
    float r = length(camera);
    float rmu = dot(camera, view_ray);
    float distance_to_top = -rmu - sqrtf(rmu * rmu - r * r + a.top_radius * a.top_radius);
    if (distance_to_top > 0.0) {
        camera = camera + view_ray * distance_to_top;
        r = a.top_radius;
        rmu += distance_to_top;
    } else if (r > a.top_radius) {
        transmittance = mk3(1.0f);
        return mk3(0.0f);
    }
    float mu = rmu / r;
    float mu_s = dot(camera, sun_direction) / r;
    float nu = dot(view_ray, sun_direction);
    bool hits_ground = RayIntersectsGround(a, r, mu);
    transmittance = hits_ground ? mk3(0.0f) : GetTransmittanceToTopAtmosphereBoundary(a, r, mu);
    f3 single_mie_scattering;
    f3 scattering;
    if (shadow_length == 0.0) {
        scattering = GetCombinedScattering(a, r, mu, mu_s, nu, hits_ground, single_mie_scattering);
    } else {
        float d = shadow_length;
        float r_p = ClampRadius(a, (float)std::sqrt(d * d + 2.0 * r * mu * d + r * r));
        float mu_p = (r * mu + d) / r_p;
        float mu_s_p = (r * mu_s + d * nu) / r_p;
        scattering = GetCombinedScattering(a, r_p, mu_p, mu_s_p, nu, hits_ground, single_mie_scattering);
        f3 shadow_transmittance = GetTransmittance(a, r, mu, shadow_length, hits_ground);
        scattering = scattering * shadow_transmittance;
        single_mie_scattering = single_mie_scattering * shadow_transmittance;
    }
    f3 sky_radiance = scattering * RayleighPhaseFunction(nu) + single_mie_scattering * MiePhaseFunction(a.mie_phase_function_g, nu);
    if (a.use_luminance != 0) sky_radiance *= to_f3(a.sky_spectral_radiance_to_luminance);
    return sky_radiance;
}
// GetSkyRadianceToPoint :749-810
static f3 GetSkyRadianceToPoint(const Atm& a, f3 camera, f3 point, float shadow_length, f3 sun_direction, f3& transmittance) {
    f3 view_ray = normalize(point - camera);
    float r = length(camera);
    float rmu = dot(camera, view_ray);
    float distance_to_top = -rmu - sqrtf(rmu * rmu - r * r + a.top_radius * a.top_radius);
    if (distance_to_top > 0.0) {
        camera = camera + view_ray * distance_to_top;
        r = a.top_radius;
        rmu += distance_to_top;
    }
    float mu = rmu / r;
    float mu_s = dot(camera, sun_direction) / r;
    float nu = dot(view_ray, sun_direction);
    float d = length(point - camera);
    bool hits_ground = RayIntersectsGround(a, r, mu);
    transmittance = GetTransmittance(a, r, mu, d, hits_ground);
    f3 single_mie_scattering;
    f3 scattering = GetCombinedScattering(a, r, mu, mu_s, nu, hits_ground, single_mie_scattering);
    d = fmaxf_(d - shadow_length, 0.0f);
    float r_p = ClampRadius(a, (float)std::sqrt(d * d + 2.0 * r * mu * d + r * r));
    float mu_p = (r * mu + d) / r_p;
    float mu_s_p = (r * mu_s + d * nu) / r_p;
    f3 single_mie_scattering_p;
    f3 scattering_p = GetCombinedScattering(a, r_p, mu_p, mu_s_p, nu, hits_ground, single_mie_scattering_p);
    f3 shadow_transmittance = transmittance;
    if (shadow_length > 0.0) shadow_transmittance = GetTransmittance(a, r, mu, d, hits_ground);
    scattering = scattering - shadow_transmittance * scattering_p;
    single_mie_scattering = single_mie_scattering - shadow_transmittance * single_mie_scattering_p;
    single_mie_scattering = single_mie_scattering * smoothstep(0.0f, 0.01f, mu_s);
    f3 sky_radiance = scattering * RayleighPhaseFunction(nu) + single_mie_scattering * MiePhaseFunction(a.mie_phase_function_g, nu);
    if (a.use_luminance != 0) sky_radiance *= to_f3(a.sky_spectral_radiance_to_luminance);
    return sky_radiance;
}
// GetSunAndSkyIrradiance :812-828
static f3 GetSunAndSkyIrradiance(const Atm& a, f3 point, f3 normal, f3 sun_direction, f3& sky_irradiance) {
    float r = length(point);
    float mu_s = dot(point, sun_direction) / r;
    sky_irradiance = GetIrradiance(a, r, mu_s) * (float)((1.0 + dot(normal, point) / r) * 0.5);
    f3 sun_irradiance = to_f3(a.solar_irradiance) * GetTransmittanceToSun(a, r, mu_s) *
                        (float)std::max((double)dot(normal, sun_direction), 0.0);
    if (a.use_luminance != 0) {
        sky_irradiance *= to_f3(a.sky_spectral_radiance_to_luminance);
        sun_irradiance *= to_f3(a.sun_spectral_radiance_to_luminance);
    }
    return sun_irradiance;
}
// GetSolarRadiance :830-835
static f3 GetSolarRadiance(const Atm& a) {
    f3 solar_radiance = to_f3(a.solar_irradiance) / (kPi * a.sun_angular_radius * a.sun_angular_radius);
    if (a.use_luminance != 0) solar_radiance *= to_f3(a.sun_spectral_radiance_to_luminance);
    return solar_radiance;
}
// sample_atmosphere :839-895
static f3 sample_atmosphere(const vpt_kernel_params& kp, const Atm& a, f3 ray_pos, f3 ray_dir) {
    f3 earth_center = mk3(.0f, -a.bottom_radius, .0f);
    f3 sun_direction = degree_to_cartesian(kp.azimuth, kp.elevation);
    f3 p = ray_pos - earth_center;
    float p_dot_v = dot(p, ray_dir);
    float p_dot_p = dot(p, p);
    float ray_earth_center_squared_distance = p_dot_p - p_dot_v * p_dot_v;
    float distance_to_intersection = -p_dot_v - sqrtf(earth_center.y * earth_center.y - ray_earth_center_squared_distance);
    float ground_alpha = 0.0;
    f3 ground_radiance = mk3(0.0f);
    if (distance_to_intersection > 0.0) {
        f3 point = ray_pos + ray_dir * distance_to_intersection;
        f3 normal = normalize(point - earth_center);
        f3 sky_irradiance;
        f3 sun_irradiance = GetSunAndSkyIrradiance(a, point - earth_center, normal, sun_direction, sky_irradiance);
        ground_radiance = to_f3(a.ground_albedo) * (float)(1.0 / kPi) * (sun_irradiance + sky_irradiance);
        f3 transmittance;
        f3 in_scatter = GetSkyRadianceToPoint(a, ray_pos - earth_center, point - earth_center, .0f, sun_direction, transmittance);
        ground_radiance = ground_radiance * transmittance + in_scatter;
        ground_alpha = 1.0;
    }
    f3 transmittance_sky;
    f3 radiance_sky = GetSkyRadiance(a, ray_pos - earth_center, ray_dir, .0f, sun_direction, transmittance_sky);
    f2 sun_size = {std::tan(a.sun_angular_radius), std::cos(a.sun_angular_radius)};
    if (dot(ray_dir, sun_direction) > sun_size.y) {
        radiance_sky = radiance_sky + transmittance_sky * GetSolarRadiance(a);
    }
    ground_radiance = lerp3(radiance_sky, ground_radiance, ground_alpha);
    f3 exposure = a.use_luminance == 0 ? mk3(a.exposure) : mk3(a.exposure) * (float)1e-5;
    f3 e = -ground_radiance / to_f3(a.white_point) * exposure;
    f3 one_minus = mk3(1.0f) - mk3(std::exp(e.x), std::exp(e.y), std::exp(e.z));
    const float g = (float)(1.0 / 2.2);
    return mk3(std::pow(one_minus.x, g), std::pow(one_minus.y, g), std::pow(one_minus.z, g));
}
static inline bool has_luts(const Atm& a) {
    return a.transmittance_texture && a.scattering_texture && a.irradiance_texture && a.single_mie_scattering_texture;
}
// sample_env_tex :897-907
static f3 sample_env_tex(const vpt_kernel_params& kp, f3 wi) {
    f4 t = tex2_4(kp.env_tex, std::atan2(wi.z, wi.x) * (float)(0.5 / (double)kPi) + 0.5f,
                  std::acos(fmaxf_(fminf_(wi.y, 1.0f), -1.0f)) * (float)(1.0 / (double)kPi));
    return xyz(t);
}

// ------------------------------------------------------------------------------------
// Volume look-ups (render_kernel.cu:909-1014)
// ------------------------------------------------------------------------------------
static inline bool to_unit(const Volume& v, f3& pos) {
    pos = mat4_transform_point(v.w2i, pos);                 // world -> index space
    pos -= to_f3(v.vdb.vdb_info.bmin);
    pos.x /= (float)v.vdb.vdb_info.dim.x;
    pos.y /= (float)v.vdb.vdb_info.dim.y;
    pos.z /= (float)v.vdb.vdb_info.dim.z;
    return !(pos.x < .0f || pos.y < .0f || pos.z < .0f || pos.x > 1.0f || pos.y > 1.0f || pos.z > 1.0f);
}
// get_density :984-1001
static inline float get_density(Ctx& c, f3 pos, const Volume& v) {
    c.st.density_lookups++;
    if (!to_unit(v, pos)) return .0f;
    return tex_sample(as_tex(v.vdb.vdb_info.density_texture), pos.x, pos.y, pos.z, true).x;
}
// sum_density :1003-1014
static inline float sum_density(Ctx& c, f3 ray_pos, const Node* leaf) {
    float density = 0.0f;
    for (int i = 0; i < leaf->num_volumes; ++i) density += get_density(c, ray_pos, c.sc->vols[leaf->vol_indices[i]]);
    return density;
}
// get_color :909-929
static inline f3 get_color(Ctx& c, f3 pos, const Volume& v) {
    if (!v.vdb.vdb_info.has_color) return WHITE();
    c.st.color_lookups++;
    if (!to_unit(v, pos)) return mk3(.0f);
    return xyz(tex_sample(as_tex(v.vdb.vdb_info.color_texture), pos.x, pos.y, pos.z, true));
}
// sum_color :931-943 (component-wise max)
static inline f3 sum_color(Ctx& c, f3 ray_pos, const Node* leaf) {
    f3 color = mk3(0.0f);
    for (int i = 0; i < leaf->num_volumes; ++i) color = fmax3(color, get_color(c, ray_pos, c.sc->vols[leaf->vol_indices[i]]));
    return color;
}
// get_emission :945-968
static inline f3 get_emission(Ctx& c, f3 pos, const Volume& v) {
    if (!v.vdb.vdb_info.has_emission) return BLACK();
    c.st.emission_lookups++;
    if (!to_unit(v, pos)) return mk3(.0f);
    float index = tex_sample(as_tex(v.vdb.vdb_info.emission_texture), pos.x, pos.y, pos.z, true).x;
    index = clampf(index * 255.0f / c.kp->emission_pivot, .0f, 255.0f);
    return to_f3(c.kp->emission_texture[(int)index]) * c.kp->emission_scale;
}
// sum_emission :970-982
static inline f3 sum_emission(Ctx& c, f3 ray_pos, const Node* leaf) {
    f3 e = mk3(0.0f);
    for (int i = 0; i < leaf->num_volumes; ++i) e += get_emission(c, ray_pos, c.sc->vols[leaf->vol_indices[i]]);
    return e;
}

// get_quadrant :1102-1115
static inline int get_quadrant(const Node* n, f3 pos) {
    int child_idx = -1;
    for (int i = 0; i < 8; ++i) {
        if (n->has_children) {
            if (contains(n->children[i]->bbox, pos)) {
                child_idx = i;
                break;
            }
        }
    }
    return child_idx;
}

// get_closest_object :1118-1135
static inline int get_closest_object(Ctx& c, f3 ray_pos, f3 ray_dir, float& t_min) {
    float tmin1 = kInf, tmax1 = -kInf, tmin2 = kInf, tmax2 = -kInf;
    bool i1 = box_intersect(c.sc->root->bbox, ray_pos, ray_dir, tmin1, tmax1);
    bool i2 = sphere_intersect(*c.sphere, ray_pos, ray_dir, tmin2, tmax2) != 0;
    if (i1 && !i2) { t_min = tmin1; return 1; }
    if (!i1 && i2) { t_min = tmin2; return 2; }
    if (i1 && i2) {
        if (tmin1 < tmin2) { t_min = tmin1; return 1; }
        if (tmin2 < tmin1) { t_min = tmin2; return 2; }
    }
    return 0;
}

// The three-level point location + empty-node push shared by sample/Tr/estimate_emission
// (render_kernel.cu:1193-1227, 1293-1327, 1609-1643).  Returns:
//   0 -> `leaf` is a non-empty leaf containing ray_pos,  1 -> pushed, caller `continue`s,
//   2 -> ray_pos is outside the tree, caller `break`s.
static inline int locate(Ctx& c, f3& ray_pos, f3 ray_dir, float& t_min, float& t_max, const Node*& leaf) {
    const Node* n = c.sc->root;
    for (int level = 0; level < 3; ++level) {
        int q = get_quadrant(n, ray_pos);
        if (q > -1) {
            const Node* ch = n->children[q];
            if (ch->num_volumes == 0) {
                box_intersect(ch->bbox, ray_pos, ray_dir, t_min, t_max);
                t_max = fmaxf_(t_max, 0.1f);
                ray_pos += ray_dir * t_max;
                c.st.skip_steps++;
                return 1;
            }
            n = ch;
        } else {
            return 2;
        }
    }
    leaf = n;
    return 0;
}

// Tr :1138-1273  (RESIDUAL_RATIO_TRACKING + DDA_STEP_TRUE)
static f3 Tr(Ctx& c, Rng& rng, f3 ray_pos, f3 ray_dir) {
    const Node* root = c.sc->root;
    f3 tr = WHITE();
    float t_min, t_max, geo_dist = .0f, distance = .0f, t = 0.0f;
    if (!contains(root->bbox, ray_pos)) {
        if (box_intersect(root->bbox, ray_pos, ray_dir, t_min, t_max)) ray_pos += ray_dir * (t_min + EPS);
        else return tr;
    }
    box_intersect(root->bbox, ray_pos, ray_dir, t_min, distance);
    if (sphere_intersect(*c.sphere, ray_pos, ray_dir, geo_dist, t_max)) return BLACK();

    float sigma_c = root->min_extinction;
    float sigma_r_inv = 1.0f / (root->max_extinction - sigma_c);
    float T_c = std::exp(-sigma_c * distance);

    while (true) {
        const Node* leaf = nullptr;
        int s = locate(c, ray_pos, ray_dir, t_min, t_max, leaf);
        if (s == 1) continue;
        if (s == 2) break;

        t -= orc_logf(1 - rnd(&rng)) * sigma_r_inv * c.kp->tr_depth;
        c.st.tracking_steps++;
        if (t >= distance) break;
        ray_pos += ray_dir * t;                              // cumulative t (Q-list 1)
        if (!contains(root->bbox, ray_pos)) break;
        float density = sum_density(c, ray_pos, leaf);
        tr *= 1 - ((density - sigma_c) * sigma_r_inv);
        if (length(tr) < EPS) break;
    }
    return clamp3(tr * T_c, .0f, 1.0f);
}

// estimate_emission :1275-1339
static f3 estimate_emission(Ctx& c, Rng& rng, f3 ray_pos, f3 ray_dir) {
    if (c.kp->emission_scale == 0) return BLACK();
    const Node* root = c.sc->root;
    f3 emission = BLACK();
    float t_min, t_max, t = 0.0f;
    while (true) {
        const Node* leaf = nullptr;
        int s = locate(c, ray_pos, ray_dir, t_min, t_max, leaf);
        if (s == 1) continue;
        if (s == 2) break;
        float inv_max_density = 1 / root->max_extinction;
        t -= orc_logf(1 - rnd(&rng)) * inv_max_density * c.kp->tr_depth / c.kp->extinction.x;
        c.st.tracking_steps++;
        ray_pos += ray_dir * t;
        if (!contains(root->bbox, ray_pos)) break;
        emission += sum_emission(c, ray_pos, leaf);
    }
    return emission;
}

// sample :1556-1681 (DDA_STEP_TRUE branch)
static f3 sample(Ctx& c, Rng& rng, f3& ray_pos, const f3& ray_dir, bool& interaction, int& obj, float& Alpha) {
    const Node* root = c.sc->root;
    float t_min, t_max, geo_dist = .0f, distance = .0f, t = 0.0f;
    bool geo = false;   // uninitialised in the reference (:1572); false is the only sane reading
    while (true) {
        const Node* leaf = nullptr;
        int s = locate(c, ray_pos, ray_dir, t_min, t_max, leaf);
        if (s == 1) continue;
        if (s == 2) break;

        float inv_max_density = 1.0f / root->max_extinction;
        float inv_density_mult = 1.0f / c.kp->density_mult;
        box_intersect(root->bbox, ray_pos, ray_dir, t_min, distance);
        if (sphere_intersect(*c.sphere, ray_pos, ray_dir, geo_dist, t_max)) {
            distance = geo_dist;
            geo = true;
        }
        t -= orc_logf(1 - rnd(&rng)) * inv_max_density * inv_density_mult;
        c.st.tracking_steps++;
        if (t >= distance) {
            if (geo) obj = 2;
            break;
        }
        ray_pos += ray_dir * t;
        if (!contains(root->bbox, ray_pos)) break;

        float density = sum_density(c, ray_pos, leaf);
        f3 Cd = sum_color(c, ray_pos, leaf);
        int index = (int)std::floor(fminf_(fmaxf_((density * inv_max_density * 255.0f / c.kp->emission_pivot), 0.0f), 255.0f));
        f3 density_color = to_f3(c.kp->density_color_texture[index]);
        if (Alpha < 1.0f) Alpha += density;
        if (density * inv_max_density > rnd(&rng)) {
            interaction = true;
            return (to_f3(c.kp->albedo) * Cd * density_color / to_f3(c.kp->extinction)) * (float)c.kp->energy_inject;
        }
    }
    return WHITE();
}

// ------------------------------------------------------------------------------------
// Environment sampling (integrator != 0 only), render_kernel.cu:167-269, 1342-1443
// ------------------------------------------------------------------------------------
// draw_sample_from_distribution :167-253 (rng BY VALUE)
static float draw_sample_from_distribution(const vpt_kernel_params& kp, Rng rng, f3& wo) {
    float xi = rnd(&rng);
    float zeta = rnd(&rng);
    float pdf = 1.0f;
    int v = 0;
    int res = kp.env_sample_tex_res;
    int first = 0, len = res;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (tex1(kp.env_marginal_cdf_tex, (float)middle) <= xi) {
            first = middle + 1;
            len -= half + 1;
        } else len = half;
    }
    v = clampi(first - 1, 0, res - 2);
    float dv = xi - tex1(kp.env_marginal_cdf_tex, (float)v);
    float d_cdf_marginal = tex1(kp.env_marginal_cdf_tex, (float)(v + 1)) - tex1(kp.env_marginal_cdf_tex, (float)v);
    if (d_cdf_marginal > .0f) dv /= d_cdf_marginal;
    float marginal_pdf = tex1(kp.env_marginal_func_tex, v + dv) / kp.env_marginal_int;
    float theta = (((float)v + dv) / (float)res) * kPi;
    int u;
    first = 0, len = res;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (tex2(kp.env_cdf_tex, (float)middle, (float)v) <= zeta) {
            first = middle + 1;
            len -= half + 1;
        } else len = half;
    }
    u = clampi(first - 1, 0, res - 2);
    float du = zeta - tex2(kp.env_cdf_tex, (float)u, (float)v);
    float d_cdf_conditional = tex2(kp.env_cdf_tex, (float)(u + 1), (float)v) - tex2(kp.env_cdf_tex, (float)u, (float)v);
    if (d_cdf_conditional > 0) du /= d_cdf_conditional;
    float conditional_pdf = tex2(kp.env_func_tex, u + du, (float)v) / tex1(kp.env_marginal_func_tex, (float)v);
    float phi = (((float)u + du) / (float)res) * kPi * 2.0f;
    float cos_theta = orc_cosf(theta);
    float sin_theta = orc_sinf(theta);
    float sin_phi = orc_sinf(phi);
    float cos_phi = orc_cosf(phi);
    wo = normalize(mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
    pdf = (marginal_pdf * conditional_pdf) / (2 * kPi * kPi * sin_theta);
    return pdf;
}
// draw_pdf_from_distribution :258-269
static float draw_pdf_from_distribution(const vpt_kernel_params& kp, f2 point) {
    int res = kp.env_sample_tex_res;
    int iu = clampi((int)(point.x * res), 0, res - 1);
    int iv = clampi((int)(point.y * res), 0, res - 1);
    float conditional = tex2(kp.env_func_tex, (float)iu, (float)iv);
    float marginal = tex1(kp.env_marginal_func_tex, (float)iv);
    return conditional / marginal;
}
// pdf_li :1342-1354
static float pdf_li(const vpt_kernel_params& kp, f3 wi) {
    float theta = std::acos(clampf(wi.y, -1.0f, 1.0f));
    float phi = std::atan2(wi.z, wi.x);
    float sin_theta = std::sin(theta);
    if (sin_theta == .0f) return .0f;
    // INV_2_PI / INV_PI are unparenthesised macros (:85-87, Q-list 12)
    float denom = 2.0f * kPi * kPi * sin_theta;
    f2 polar_pos = {(phi * 1.0f / (2.0f * kPi)) / denom, (theta * 1.0f / kPi) / denom};
    return draw_pdf_from_distribution(kp, polar_pos);
}

// estimate_sky :1356-1443
static f3 estimate_sky(Ctx& c, Rng& rng, const f3& ray_pos, f3& ray_dir) {
    const vpt_kernel_params& kp = *c.kp;
    f3 Ld = BLACK();
    for (int i = 0; i < 1; i++) {
        f3 Li = BLACK();
        f3 wi;
        float light_pdf = .0f, phase_pdf = .0f;
        float az = rnd(&rng) * 360.0f;
        float el = rnd(&rng) * 180.0f;
        (void)az; (void)el;
        if (kp.environment_type == 0) {
            light_pdf = draw_sample_from_distribution(kp, rng, wi);
            Li = sample_atmosphere(kp, *c.atm, ray_pos, wi);
        } else {
            light_pdf = sample_spherical(rng, wi);
            Li = sample_env_tex(kp, wi);
        }
        if (light_pdf > .0f && !is_black(Li)) {
            float cos_theta = dot(ray_dir, wi);
            phase_pdf = henyey_greenstein(cos_theta, kp.phase_g1);
            if (phase_pdf > .0f) {
                f3 tr = Tr(c, rng, ray_pos, wi);
                Li *= tr;
                if (!is_black(Li)) {
                    float weight = power_heuristic(1, light_pdf, 1, phase_pdf);
                    Ld += Li * phase_pdf * weight / light_pdf;
                }
            }
        }
        wi = ray_dir;
        phase_pdf = sample_hg(wi, rng, kp.phase_g1);
        if (phase_pdf > .0f) {
            Li = BLACK();
            float weight = 1.0f;
            if (kp.environment_type == 0) light_pdf = pdf_li(kp, wi);
            else light_pdf = isotropic();
            if (light_pdf == 0.0f) return Ld;
            weight = power_heuristic(1, phase_pdf, 1, light_pdf);
            f3 tr = Tr(c, rng, ray_pos, wi);
            if (kp.environment_type == 0) Li = sample_atmosphere(kp, *c.atm, ray_pos, wi);
            else Li = sample_env_tex(kp, wi);
            if (!is_black(Li)) Ld += Li * tr * weight;
        }
    }
    return Ld;
}

// point_light::Le, light.h:104-121 (only the part before the first `return`)
static f3 point_light_Le(const vpt_point_light& l, f3 ray_pos, f3 ray_dir, float phase_g1, f3 tr) {
    f3 pos = to_f3(l.pos);
    f3 wi = normalize(pos - ray_pos);
    float cos_theta = dot(ray_dir, wi);
    float phase_pdf = henyey_greenstein(cos_theta, phase_g1);
    float sqr_dist = length(pos * pos - ray_pos * ray_pos);   // component-wise squares (Q-list, K13)
    float falloff = 1 / sqr_dist;
    return to_f3(l.color) * l.power * tr * phase_pdf * falloff;
}

// estimate_point_light :1445-1475
static f3 estimate_point_light(Ctx& c, Rng& rng, const f3& ray_pos, f3& ray_dir) {
    f3 Ld = mk3(.0f);
    int light_budget = 10;
    const int nl = (int)c.lights->num_lights;
    while (light_budget >= 0) {
        int light_index = (int)std::floor(rnd(&rng) * nl);
        // rand() is (0,1]: the reference reads light_ptr[num_lights] (out of bounds) when
        // it returns exactly 1.0f; clamp instead of restating undefined behaviour.
        if (light_index > nl - 1) light_index = nl - 1;
        const vpt_point_light& l = c.lights->light_ptr[light_index];
        f3 dir = normalize(to_f3(l.pos) - ray_pos);
        f3 tr = Tr(c, rng, ray_pos, dir);
        if (light_budget < nl) Ld += point_light_Le(l, ray_pos, ray_dir, c.kp->phase_g1, tr);
        light_budget--;
    }
    return Ld;
}

// estimate_sun :1478-1516 (the irradiance look-ups at :1504-1505 are dead code)
static f3 estimate_sun(Ctx& c, Rng& rng, const f3& ray_pos, f3& ray_dir) {
    f3 wi = degree_to_cartesian(c.kp->azimuth, c.kp->elevation);
    float cos_theta = dot(ray_dir, wi);
    float phase_pdf = henyey_greenstein(cos_theta, c.kp->phase_g1);
    f3 tr = Tr(c, rng, ray_pos, wi);
    f3 Ld = tr * phase_pdf;
    return Ld * to_f3(c.kp->sun_color) * c.kp->sun_mult;
}

// uniform_sample_one_light :1519-1554
static f3 uniform_sample_one_light(Ctx& c, const f3& ray_pos, f3& ray_dir, Rng& rng) {
    int nLights = 3;
    float light_num = rnd(&rng) * nLights;
    f3 L = BLACK();
    if (light_num < 1) {
        if (c.kp->sun_mult > .0f) L += estimate_sun(c, rng, ray_pos, ray_dir);
    } else if (light_num >= 1 && light_num < 2) {
        if (c.lights->num_lights > 0) L += estimate_point_light(c, rng, ray_pos, ray_dir);
    } else {
        if (c.kp->sky_mult > .0f) L += estimate_sky(c, rng, ray_pos, ray_dir) * c.kp->sky_mult;
    }
    return L * (float)nLights;
}

// ------------------------------------------------------------------------------------
// Integrators
// ------------------------------------------------------------------------------------
// vol_integrator :1712-1756 (rng BY VALUE)
static f3 vol_integrator(Ctx& c, Rng rng, f3 ray_pos, f3 ray_dir, float& tr) {
    const vpt_kernel_params& kp = *c.kp;
    f3 L = BLACK();
    f3 beta = WHITE();
    f3 env_pos = ray_pos;
    bool mi;
    float t, tmax;
    int obj = 0;
    if (box_intersect(c.sc->root->bbox, ray_pos, ray_dir, t, tmax)) {
        ray_pos += ray_dir * (t + EPS);
        for (int depth = 1; depth <= kp.ray_depth; depth++) {
            mi = false;
            beta *= sample(c, rng, ray_pos, ray_dir, mi, obj, tr);
            if (is_black(beta)) break;
            if (mi) {
                f3 a = beta * uniform_sample_one_light(c, ray_pos, ray_dir, rng);
                f3 b = estimate_emission(c, rng, ray_pos, ray_dir);
                L += a + b;
                sample_hg(ray_dir, rng, kp.phase_g1);
            }
        }
        ray_dir = normalize(ray_dir);
    }
    if (length(beta) > 0.9999f) ray_pos = env_pos;
    L += beta * sample_atmosphere(kp, *c.atm, ray_pos, ray_dir);
    tr = fminf_(tr, 1.0f);
    return L;
}

// direct_integrator :1760-1857 (rng BY VALUE)
static f3 direct_integrator(Ctx& c, Rng rng, f3 ray_pos, f3 ray_dir, float& tr) {
    const vpt_kernel_params& kp = *c.kp;
    f3 L = BLACK();
    f3 beta = WHITE();
    bool mi = false;
    f3 env_pos = ray_pos;
    float t_min;
    int obj;
    for (int ray_depth = 1; ray_depth <= kp.ray_depth; ray_depth++) {
        obj = get_closest_object(c, ray_pos, ray_dir, t_min);
        if (obj == 1) {
            ray_pos += ray_dir * (t_min + EPS);
            for (int volume_depth = 1; volume_depth <= kp.volume_depth; volume_depth++) {
                mi = false;
                beta *= sample(c, rng, ray_pos, ray_dir, mi, obj, tr);
                if (is_black(beta) || obj == 2) break;
                if (mi) sample_hg(ray_dir, rng, kp.phase_g1);
            }
            if (mi) {
                L += estimate_sun(c, rng, ray_pos, ray_dir) * beta;
                if (c.lights->num_lights > 0) L += estimate_point_light(c, rng, ray_pos, ray_dir) * beta;
            }
            if (kp.emission_scale > 0 && mi) L += estimate_emission(c, rng, ray_pos, ray_dir);
        }
        obj = get_closest_object(c, ray_pos, ray_dir, t_min);
        if (obj == 2) {
            const vpt_sphere& s = *c.sphere;
            ray_pos += ray_dir * t_min;
            f3 normal = normalize((ray_pos - to_f3(s.center)) / s.radius);
            f3 nl = dot(normal, ray_dir) < 0 ? normal : normal * -1;
            float phi = 2 * kPi * rnd(&rng);
            float r2 = rnd(&rng);
            float r2s = sqrtf(r2);
            f3 w = normalize(nl);
            f3 u = normalize(cross(((double)std::fabs(w.x) > .1 ? mk3(0, 1, 0) : mk3(1, 0, 0)), w));
            f3 v = cross(w, u);
            f3 hemisphere_dir = normalize(u * orc_cosf(phi) * r2s + v * orc_sinf(phi) * r2s + w * sqrtf(1 - r2));
            f3 ref = reflect(ray_dir, nl);
            ray_dir = lerp3(ref, hemisphere_dir, s.roughness);
            f3 light_dir = degree_to_cartesian(kp.azimuth, kp.elevation);
            ray_pos += normal * EPS;
            beta *= to_f3(s.color);
            f3 v_tr = Tr(c, rng, ray_pos, light_dir);
            L += to_f3(kp.sun_color) * kp.sun_mult * v_tr * fmaxf_(dot(light_dir, normal), .0f) * beta;
            env_pos = ray_pos;
        }
    }
    if (kp.environment_type == 0) {
        // without bound look-up tables the sky term is skipped; orc_render only allows that
        // when sky_mult == 0, where the reference's term is 0 as well (finite sky * 0)
        if (has_luts(*c.atm)) L += sample_atmosphere(kp, *c.atm, env_pos, ray_dir) * beta * kp.sky_mult * to_f3(kp.sky_color);
    } else {
        f3 t = sample_env_tex(kp, ray_dir);
        L += t * to_f3(kp.sky_color) * beta * isotropic();
    }
    tr = fminf_(tr, 1.0f);
    return L;
}

// depth_calculator :1859-1889 (rng BY VALUE)
static float depth_calculator(Ctx& c, Rng rng, f3 ray_pos, f3 ray_dir, float& tr) {
    f3 orig = ray_pos;
    bool mi = false;
    float t_min;
    int obj = get_closest_object(c, ray_pos, ray_dir, t_min);
    if (obj == 1) {
        ray_pos += ray_dir * (t_min + EPS);
        sample(c, rng, ray_pos, ray_dir, mi, obj, tr);
        if (mi) return length(orig - ray_pos);
        else return .0f;
    }
    if (obj == 2) {
        ray_pos += ray_dir * t_min;
        return length(orig - ray_pos);
    }
    return .0f;
}

// rtt_and_odt_fit :2208-2213
static inline f3 rtt_and_odt_fit(f3 v) {
    f3 a = v * (v + 0.0245786f) - 0.000090537f;
    f3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
static inline f3 mat3_mul(const float m[9], f3 v) {          // matrix_math.h mat3 * float3 (row-major ctor)
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
               m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

struct PixelOut { f3 value; float tr; float depth; uint64_t draws; };

// the per-thread body of volume_rt_kernel up to (not including) accumulation, :2227-2274
static PixelOut trace_pixel(Ctx& c, const vpt_camera& cam, int x, int y, const vpt_float3* bn_snapshot) {
    const vpt_kernel_params& kp = *c.kp;
    const unsigned int idx = y * kp.resolution.x + x;
    Rng rng;
    rng_init(&rng, idx, (uint64_t)kp.iteration * 4096);
    int bn_index = (y % 256) * 256 + (x % 256);
    vpt_float3 bn = bn_snapshot[bn_index];
    float u = (float)(x + bn.x) / (float)kp.resolution.x;
    float v = (float)(y + bn.y) / (float)kp.resolution.y;
    f3 A, B;
    get_ray(cam, u, v, &rng, A, B);
    f3 ray_dir = normalize(B);
    f3 ray_pos = A;
    PixelOut o;
    o.value = WHITE();
    o.depth = .0f;
    o.tr = .0f;
    if (kp.iteration < kp.max_interactions && kp.render) {
        o.depth = depth_calculator(c, rng, ray_pos, ray_dir, o.tr);
        if (kp.integrator) o.value = vol_integrator(c, rng, ray_pos, ray_dir, o.tr);
        else o.value = direct_integrator(c, rng, ray_pos, ray_dir, o.tr);
    }
    o.draws = rng.draws;
    return o;
}

}  // namespace

// ======================================================================================
// C API
// ======================================================================================
extern "C" {

// 32 (default: the parity contract) or 8 (diagnostic model of the CUDA texture unit's weights, volume grids only); returns the previous value
int orc_set_volume_tex_weights(int bits) {
    const int prev = g_volume_weight_bits;
    if (bits == 8 || bits == 32) g_volume_weight_bits = bits;
    return prev;
}

vpt_texture_t orc_texture_create(const vpt_texture_desc* desc, const float* data) {
    Tex* t = new Tex();
    t->d = *desc;
    if (t->d.height < 1) t->d.height = 1;
    if (t->d.depth < 1) t->d.depth = 1;
    t->data = data;
    return (vpt_texture_t)(uintptr_t)t;
}
void orc_texture_destroy(vpt_texture_t tex) { delete reinterpret_cast<Tex*>((uintptr_t)tex); }
void orc_texture_sample(vpt_texture_t tex, float u, float v, float w, float out[4]) {
    f4 r = tex_sample(as_tex(tex), u, v, w);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void orc_philox4x32_10(const unsigned int ctr[4], const unsigned int key[2], unsigned int out[4]) {
    philox10(ctr, key, out);
}
void orc_curand_uniform_stream(unsigned long long seed, unsigned long long offset, int n, float* out) {
    Rng r;
    rng_init(&r, seed, offset);
    for (int i = 0; i < n; ++i) out[i] = rnd(&r);
}
float orc_det_logf(float x) { return orc_logf(x); }
float orc_det_sinf(float x) { return orc_sinf(x); }
float orc_det_cosf(float x) { return orc_cosf(x); }

int orc_octree_info_get(const vpt_gpu_vdb* volumes, int num_volumes, orc_octree_info* out) {
    if (!volumes || num_volumes <= 0 || !out) return VPT_E_INVALID;
    Scene sc;
    sc.build(volumes, num_volumes);
    out->root_pmin = to_v3(sc.root->bbox.pmin);
    out->root_pmax = to_v3(sc.root->bbox.pmax);
    out->max_extinction = sc.root->max_extinction;
    out->min_extinction = sc.root->min_extinction;
    out->nonempty[0] = out->nonempty[1] = out->nonempty[2] = 0;
    out->total_nodes = (int)sc.pool.size();
    for (auto& n : sc.pool) {
        if (n.get() == sc.root) continue;
        if (n->num_volumes > 0) out->nonempty[3 - n->depth]++;
    }
    return 0;
}

int orc_octree_locate(const vpt_gpu_vdb* volumes, int num_volumes, vpt_float3 p, int* nvol) {
    Scene sc;
    sc.build(volumes, num_volumes);
    const Node* n = sc.root;
    int path = 0;
    for (int level = 0; level < 3; ++level) {
        int q = get_quadrant(n, to_f3(p));
        if (q < 0) return -1;
        path = path * 8 + q;
        n = n->children[q];
        if (n->num_volumes == 0) { if (nvol) *nvol = 0; return -(100 + level); }
    }
    if (nvol) *nvol = n->num_volumes;
    return path;
}

float orc_density_at(const vpt_gpu_vdb* volumes, int num_volumes, vpt_float3 p) {
    Scene sc;
    sc.build(volumes, num_volumes);
    Ctx c = {};
    c.sc = &sc;
    float d = 0.0f;
    for (int i = 0; i < num_volumes; ++i) d += get_density(c, to_f3(p), sc.vols[i]);
    return d;
}

int orc_sample_pixel(const vpt_camera* cam, const vpt_light_list* lights, const vpt_gpu_vdb* volumes, int num_volumes,
                     const vpt_sphere* ref_sphere, const vpt_atmosphere_parameters* atmosphere,
                     const vpt_kernel_params* kp, int x, int y, float out[5]) {
    if (!cam || !lights || !volumes || !ref_sphere || !atmosphere || !kp || !out) return VPT_E_INVALID;
    Scene sc;
    sc.build(volumes, num_volumes);
    Ctx c = {};
    c.sc = &sc; c.kp = kp; c.atm = atmosphere; c.sphere = ref_sphere; c.lights = lights;
    PixelOut o = trace_pixel(c, *cam, x, y, kp->blue_noise_buffer);
    out[0] = o.value.x; out[1] = o.value.y; out[2] = o.value.z; out[3] = o.tr; out[4] = o.depth;
    return 0;
}

int orc_render(const vpt_camera* cam, const vpt_light_list* lights, const vpt_gpu_vdb* volumes, int num_volumes,
               const vpt_sphere* ref_sphere, const vpt_atmosphere_parameters* atmosphere,
               const vpt_kernel_params* kp_in, unsigned int iter_count, unsigned int iter_stride, int nthreads,
               orc_stats* stats) {
    return orc_render_subset(cam, lights, volumes, num_volumes, ref_sphere, atmosphere, kp_in, iter_count, iter_stride, nthreads, 1u, stats);
}

int orc_render_subset(const vpt_camera* cam, const vpt_light_list* lights, const vpt_gpu_vdb* volumes, int num_volumes,
                      const vpt_sphere* ref_sphere, const vpt_atmosphere_parameters* atmosphere,
                      const vpt_kernel_params* kp_in, unsigned int iter_count, unsigned int iter_stride, int nthreads,
                      unsigned int pixel_step, orc_stats* stats) {
    if (!cam || !lights || !volumes || num_volumes <= 0 || !ref_sphere || !atmosphere || !kp_in) return VPT_E_INVALID;
    if (iter_stride == 0) iter_stride = 1;
    // vol_integrator's tail is always sample_atmosphere (:1752), whatever environment_type says
    if (!has_luts(*atmosphere) && ((kp_in->environment_type == 0 && kp_in->sky_mult != 0.0f) || kp_in->integrator != 0)) return VPT_E_NOT_READY;
    Scene sc;
    sc.build(volumes, num_volumes);
    vpt_kernel_params kp = *kp_in;
    const int W = (int)kp.resolution.x, H = (int)kp.resolution.y;
    orc_stats total = {};
    std::vector<vpt_float3> bn(256 * 256);
    (void)nthreads;
#ifdef _OPENMP
    int nt = nthreads > 1 ? nthreads : 1;
#endif
    static const float aces_in[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    static const float aces_out[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};

    for (unsigned int k = 0; k < iter_count; ++k) {
        kp.iteration = kp_in->iteration + k * iter_stride;
        const unsigned int local_it = kp.iteration / iter_stride;     // running-mean index
        // all threads of a launch read the pre-update blue noise (the reference races, :2241 vs :2320)
        std::memcpy(bn.data(), kp.blue_noise_buffer, sizeof(vpt_float3) * 256 * 256);
        orc_stats iter_tot = {};
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
        {
            Ctx c = {};
            c.sc = &sc; c.kp = &kp; c.atm = atmosphere; c.sphere = ref_sphere; c.lights = lights;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
            for (int y = 0; y < H; ++y) {
                for (int x = 0; x < W; ++x) {
                    const unsigned int idx = y * W + x;
                    if (pixel_step > 1u && idx % pixel_step != 0u) continue;      // a lattice of the frame (bounded CPU samples of big frames)
                    PixelOut o = trace_pixel(c, *cam, x, y, bn.data());
                    c.st.samples++;
                    c.st.rng_draws += o.draws;
                    if (o.draws > c.st.max_draws_per_sample) c.st.max_draws_per_sample = o.draws;
                    f3 value = o.value;
                    float tr = o.tr;
                    float depth = o.depth;
                    f3 cost = BLACK();
                    // :2263-2264
                    if (is_nan3(value) || is_inf3(value)) value = to_f3(kp.accum_buffer[idx]);
                    if (std::isnan(tr) || std::isinf(tr)) tr = 1.0f;
                    // :2266-2274
                    float aof = 1 / cam->lens_radius;
                    aof = clampf(aof, .0f, FLT_MAX);
                    if (cam->viz_dof) {
                        if (depth > (cam->focus_dist + aof)) value = lerp3(value, mk3(1, 0, 0), 0.5f);
                        if (depth < (cam->focus_dist - aof)) value = lerp3(value, mk3(0, 0, 1), 0.5f);
                        if (depth > (cam->focus_dist - aof) && depth < (cam->focus_dist + aof)) value = lerp3(value, mk3(0, 1, 0), 0.5f);
                    }
                    // :2278-2287
                    if (local_it == 0) {
                        kp.accum_buffer[idx] = to_v3(value);
                        if (kp.cost_buffer) kp.cost_buffer[idx] = to_v3(cost);
                        if (kp.depth_buffer) kp.depth_buffer[idx] = depth;
                    } else if (kp.iteration < kp.max_interactions) {
                        f3 acc = to_f3(kp.accum_buffer[idx]);
                        kp.accum_buffer[idx] = to_v3(acc + (value - acc) / (float)(local_it + 1));
                        if (kp.cost_buffer) {
                            f3 cb = to_f3(kp.cost_buffer[idx]);
                            kp.cost_buffer[idx] = to_v3(cb + (cost - cb) / (float)(local_it + 1));
                        }
                        if (kp.depth_buffer) kp.depth_buffer[idx] = kp.depth_buffer[idx] + (depth - kp.depth_buffer[idx]) / (float)(local_it + 1);
                    }
                    // :2292-2316
                    f3 val = mat3_mul(aces_in, to_f3(kp.accum_buffer[idx]));
                    val = rtt_and_odt_fit(val);
                    val = mat3_mul(aces_out, val) * kp.exposure_scale;
                    const float ig = (float)(1.0 / 2.2);
                    const unsigned int r = (unsigned int)(255.0f * fminf_(std::pow(fmaxf_(val.x, 0.0f), ig), 1.0f));
                    const unsigned int g = (unsigned int)(255.0f * fminf_(std::pow(fmaxf_(val.y, 0.0f), ig), 1.0f));
                    const unsigned int b = (unsigned int)(255.0f * fminf_(std::pow(fmaxf_(val.z, 0.0f), ig), 1.0f));
                    if (kp.display_buffer) kp.display_buffer[idx] = 0xff000000 | (r << 16) | (g << 8) | b;
                    if (kp.raw_buffer) { vpt_float4 raw = {val.x, val.y, val.z, tr}; kp.raw_buffer[idx] = raw; }
                }
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            {
                iter_tot.samples += c.st.samples;
                iter_tot.density_lookups += c.st.density_lookups;
                iter_tot.color_lookups += c.st.color_lookups;
                iter_tot.emission_lookups += c.st.emission_lookups;
                iter_tot.tracking_steps += c.st.tracking_steps;
                iter_tot.skip_steps += c.st.skip_steps;
                iter_tot.rng_draws += c.st.rng_draws;
                if (c.st.max_draws_per_sample > iter_tot.max_draws_per_sample) iter_tot.max_draws_per_sample = c.st.max_draws_per_sample;
            }
        }
        total.samples += iter_tot.samples;
        total.density_lookups += iter_tot.density_lookups;
        total.color_lookups += iter_tot.color_lookups;
        total.emission_lookups += iter_tot.emission_lookups;
        total.tracking_steps += iter_tot.tracking_steps;
        total.skip_steps += iter_tot.skip_steps;
        total.rng_draws += iter_tot.rng_draws;
        if (iter_tot.max_draws_per_sample > total.max_draws_per_sample) total.max_draws_per_sample = iter_tot.max_draws_per_sample;
        // :2320-2325, applied after the launch (iter_stride launches' worth for striping)
        for (unsigned int s = 0; s < iter_stride; ++s) {
            // `if (idx < 256*256)` runs for idx < W*H only: smaller images advance just their first W*H entries
            const int live = (long long)W * H < 256 * 256 ? W * H : 256 * 256;
            for (int i = 0; i < live; ++i) {
                vpt_float3 v = kp.blue_noise_buffer[i];
                const float phi = (1.0f + sqrtf(5.0f)) / 2.0f;
                v.x = std::fmod(v.x + phi, 1.0f);
                v.y = std::fmod(v.y + phi, 1.0f);
                v.z = std::fmod(v.z + phi, 1.0f);
                kp.blue_noise_buffer[i] = v;
            }
        }
    }
    if (stats) *stats = total;
    return 0;
}

}  // extern "C"
