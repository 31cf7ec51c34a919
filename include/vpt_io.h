/*
 * vpt_io.h -- C ABI of the host-side data formats either side of the hot path:
 * what the reference's main.cpp reads before it can launch `volume_rt_kernel` and writes after.
 * Pure host code inside libvpt_hip.so (no GPU needed), no third-party libraries except zlib.
 *
 *   OpenVDB files  -> dense grids + VDB_INFO   GPU_VDB::loadVDB, source/gpu_vdb/gpu_vdb.cpp:105-472
 *                                              (OpenVDB itself is not linked: the file format
 *                                              222-224 is parsed directly)
 *   .ins files     -> instances / point lights read_instance_file, source/main.cpp:980-1102
 *   BN0.bmp        -> blue-noise float3        load_texture_bmp_gpu, source/util/fileIO.cpp:460-495
 *   256x1 EXR LUTs -> float3[256]              load_texture_exr_gpu, source/util/fileIO.cpp:356-390
 *   Radiance .hdr  -> float4 lat-long map      load_hdr_float4, source/hdr_loader.h:249-277
 *   PFM / PPM / PNG <- accum / display buffers (the reference writes EXR/PNG/JPG/TGA through
 *                                              OpenImageIO, fileIO.cpp:53-288; here the two
 *                                              dependency-free formats and PNG over zlib)
 * Every function returns VPT_OK or a negative VPT_E_* code (vpt_abi.h); vpt_io_last_error() gives
 * the message.  Buffers returned through `float **` are malloc'ed: release with vpt_io_free.
 */
#ifndef VPT_IO_H_
#define VPT_IO_H_

#include "vpt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *vpt_io_last_error(void);
void vpt_io_free(void *p);

/* ---- OpenVDB ------------------------------------------------------------------------------- */
typedef struct vpt_io_volume vpt_io_volume;
/* GPU_VDB::loadVDB(filename, density_channel, emission_channel, color_channel): reads the named
 * grids (emission / colour may be NULL or "" or absent from the file), densifies each over ITS OWN
 * active-voxel bounding box in LayoutXYZ (x fastest), computes max/min density
 * (gpu_vdb.cpp:206-207) and the index->world matrix.  Supports Tree_float_5_4_3 / Tree_vec3s_5_4_3,
 * compression ZIP | ACTIVE_MASK | BLOSC (LZ4 / zlib / memcpy codecs), the linear map types. */
int  vpt_io_vdb_load(const char *filename, const char *density_channel, const char *emission_channel,
                     const char *color_channel, vpt_io_volume **out);
void vpt_io_vdb_free(vpt_io_volume *vol);
/* VDB_INFO (texture handles zero) + xform exactly as loadVDB leaves them */
int  vpt_io_vdb_info(const vpt_io_volume *vol, vpt_gpu_vdb *out);
/* which: 0 density (f32), 1 emission (f32), 2 colour (float4, w = 1).  *data is owned by vol. */
int  vpt_io_vdb_grid(const vpt_io_volume *vol, int which, const float **data, vpt_int3 *dim);
/* statistics used by the tests: leaves, active voxels, active tiles of grid `which` */
int  vpt_io_vdb_stats(const vpt_io_volume *vol, int which, long long out[3]);
/* loadVDB + texture creation on a context: fills out->vdb_info.*_texture */
int  vpt_io_vdb_upload(vpt_ctx *ctx, const vpt_io_volume *vol, vpt_gpu_vdb *out);

/* ---- .ins instance / light files ------------------------------------------------------------- */
typedef struct vpt_io_instance {
    double position[3];
    double rotation[4];     /* quaternion x, y, z, w */
    double scale;
} vpt_io_instance;
typedef struct vpt_io_ins vpt_io_ins;
int  vpt_io_ins_read(const char *filename, vpt_io_ins **out);
void vpt_io_ins_free(vpt_io_ins *ins);
/* 1 if the file is a "light" file (main.cpp:989), else 0 */
int  vpt_io_ins_is_light_file(const vpt_io_ins *ins);
int  vpt_io_ins_num_files(const vpt_io_ins *ins);
const char *vpt_io_ins_file_name(const vpt_io_ins *ins, int file);
int  vpt_io_ins_num_instances(const vpt_io_ins *ins, int file);
const vpt_io_instance *vpt_io_ins_instances(const vpt_io_ins *ins, int file);
int  vpt_io_ins_num_lights(const vpt_io_ins *ins);
const vpt_point_light *vpt_io_ins_lights(const vpt_io_ins *ins);

/* ---- images -------------------------------------------------------------------------------------- */
/* 24-bit BMP -> float3, rows top-down, x = R/255, y = B/255, z = G/255 (fileIO.cpp:482-484) */
int  vpt_io_load_bmp(const char *filename, float **rgb, int *width, int *height);
/* scanline OpenEXR (uncompressed or ZIP/ZIPS, HALF or FLOAT channels) -> float3 RGB, alpha dropped */
int  vpt_io_load_exr_rgb(const char *filename, float **rgb, int *width, int *height);
/* Radiance RGBE (.hdr, flat or RLE) -> float4 with w = 0 (calloc'ed 4th component, hdr_loader.h:262) */
int  vpt_io_load_hdr(const char *filename, float **rgba, int *width, int *height);
/* little-endian PFM (rows bottom-up as the format wants) from a top-down float3 / float4 buffer */
int  vpt_io_write_pfm(const char *filename, const float *pixels, int channels, int width, int height);
/* binary PPM from the 0xffRRGGBB display buffer */
int  vpt_io_write_ppm(const char *filename, const unsigned int *display, int width, int height);
/* PNG, 8 bits per channel, from the 0xffRRGGBB display buffer (save_texture_png(uint32_t*), fileIO.cpp:140-154): RGB, or RGBA with the
 * word's top byte as alpha when with_alpha != 0.  Deflate and CRC-32 come from the zlib the library already links. */
int  vpt_io_write_png(const char *filename, const unsigned int *display, int width, int height, int with_alpha);
/* PNG from a top-down float3 / float4 buffer, converted as OpenImageIO converts FLOAT to UINT8 (the float3 / float4 overloads of save_texture_png,
 * fileIO.cpp:110-138): clamp to [0, 1], x 255, round to nearest; NaN -> 0 */
int  vpt_io_write_png_float(const char *filename, const float *pixels, int channels, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* VPT_IO_H_ */
