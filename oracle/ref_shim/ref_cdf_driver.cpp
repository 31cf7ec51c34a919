// TEST INFRASTRUCTURE -- oracle/_ref/ref_env_cdf: the reference's OWN environment importance-table fill, compiled for the
// CPU from the lines where they lie in /root/reference/source/main.cpp:
//     :181-312   solveQuadratic, raySphereIntersect, degree_to_radians, degree_to_cartesian, the host sample_atmosphere
//     :662-752   the table fill of create_cdf (everything between its debug log line and "// End array filling")
// main.cpp as a whole cannot be compiled here (GLFW, ImGui, OIDN, <Windows.h>, the CUDA driver API); the two line ranges are
// plain C++ over helper_math.h and Kernel_params.  oracle/Makefile extracts them with `sed -n` into oracle/_ref/*.inc
// (git-ignored, never committed) and this file #includes the extracts -- it contains no reference code itself.
// It pins the product's restatement (csrc/vpt_env.hip: vpt_env_cdf_build), which tests/test_env_cdf.py compares bit for bit.
//
// The fill reads one element BEFORE three of its arrays and writes one element past the end of marginal_cdf
// (main.cpp:690, :698 at the first texel, :729 at the first row, :750).  To give that a definite meaning without touching
// the source, this program replaces operator new[] with one that returns zeroed storage with zeroed guard space on both
// sides: the stray reads yield 0.0f -- what the product defines them to be -- and the stray writes land in the guard.
// A standalone executable (not a library), so the replaced operator new[] stays private to it.
//
//   ref_env_cdf <azimuth> <elevation> <sky_r> <sky_g> <sky_b> <out.bin>
//   out.bin: u32 res | f32 marginal_int | val[res*res*3] | func[res*res] | cdf[res*res] | marginal_func[res] | marginal_cdf[res]
#include "cuda_runtime.h"

#include <cfloat>
#include <new>

void* operator new[](std::size_t n) {
    char* p = static_cast<char*>(calloc(1, n + 256));
    if (!p) throw std::bad_alloc();
    return p + 128;
}
void operator delete[](void* p) noexcept {
    if (p) free(static_cast<char*>(p) - 128);
}
void operator delete[](void* p, std::size_t) noexcept {
    if (p) free(static_cast<char*>(p) - 128);
}

#include "helper_math.h"      // the reference's (source/common)
#include "kernel_params.h"    // the reference's (source)

#include "main_env_functions.inc"          // main.cpp:181-312, extracted by oracle/Makefile

static void fill_tables(Kernel_params& kernel_params, const char* path) {
#include "main_create_cdf_fill.inc"        // main.cpp:662-752: declares res, val, func, cdf, marginal_func, marginal_cdf, marginal_int
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); exit(1); }
    const unsigned r = res;
    fwrite(&r, 4, 1, f);
    fwrite(&marginal_int, 4, 1, f);
    fwrite(val, sizeof(float3), (size_t)res * res, f);
    fwrite(func, 4, (size_t)res * res, f);
    fwrite(cdf, 4, (size_t)res * res, f);
    fwrite(marginal_func, 4, res, f);
    fwrite(marginal_cdf, 4, res, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: ref_env_cdf <azimuth> <elevation> <sky_r> <sky_g> <sky_b> <out.bin>\n");
        return 2;
    }
    Kernel_params kp;
    memset(&kp, 0, sizeof(kp));
    kp.azimuth = (float)atof(argv[1]);
    kp.elevation = (float)atof(argv[2]);
    kp.sky_color = make_float3((float)atof(argv[3]), (float)atof(argv[4]), (float)atof(argv[5]));
    fill_tables(kp, argv[6]);
    return 0;
}
