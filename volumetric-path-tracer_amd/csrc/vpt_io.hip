// vpt_io.hip -- host-side data formats either side of the hot path (include/vpt_io.h): a direct
// parser of the OpenVDB file format (versions 222-224) that reproduces what
// GPU_VDB::loadVDB (source/gpu_vdb/gpu_vdb.cpp:105-472) extracts through the OpenVDB library, the
// `.ins` instance / light files (source/main.cpp:980-1102), the BMP / EXR / Radiance-HDR inputs
// (source/util/fileIO.cpp:356-495, source/hdr_loader.h) and PFM / PPM writers.  Host code only.
//
// Third-party formats restated from their public specifications: OpenVDB io (Archive / RootNode /
// InternalNode / LeafNode ::readTopology / readBuffers, io/Compression.h), c-blosc 1.x chunk
// layout (blosc.h / blosc.c `blosc_d`), LZ4 block format, OpenEXR scanline files, Radiance RGBE.
#include <zlib.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vpt_io.h"

namespace {

thread_local std::string g_io_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_io_error = buf;
    return code;
}

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == out.size();
}

// ---- byte reader -----------------------------------------------------------------------------
struct Reader {
    const uint8_t* b;
    size_t n, p;
    const uint8_t* take(size_t k) {
        if (p > n || k > n - p) throw ParseError("truncated file");
        const uint8_t* r = b + p;
        p += k;
        return r;
    }
    // an offset read FROM the file (grid / block / end positions): inside the buffer or a ParseError
    void seek(int64_t pos) {
        if (pos < 0 || (uint64_t)pos > (uint64_t)n) throw ParseError("offset outside the file");
        p = (size_t)pos;
    }
    template <class T>
    T get() {
        T v;
        std::memcpy(&v, take(sizeof(T)), sizeof(T));
        return v;
    }
    std::string str() {
        uint32_t len = get<uint32_t>();
        const uint8_t* s = take(len);
        return std::string((const char*)s, len);
    }
};

// ---- LZ4 block format ------------------------------------------------------------------------
size_t lz4_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    size_t ip = 0, op = 0;
    while (ip < n) {
        const unsigned token = src[ip++];
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= n) throw ParseError("lz4: truncated literal length");
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (lit > n - ip || lit > cap - op) throw ParseError("lz4: literal overrun");
        std::memcpy(dst + op, src + ip, lit);
        ip += lit;
        op += lit;
        if (ip >= n) break;                                   // last sequence has no match
        if (n - ip < 2) throw ParseError("lz4: truncated offset");
        const size_t off = src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        size_t mlen = token & 15;
        if (mlen == 15) {
            unsigned b;
            do {
                if (ip >= n) throw ParseError("lz4: truncated match length");
                b = src[ip++];
                mlen += b;
            } while (b == 255);
        }
        mlen += 4;
        if (off == 0 || off > op || mlen > cap - op) throw ParseError("lz4: bad match");
        for (size_t i = 0; i < mlen; ++i) dst[op + i] = dst[op - off + i];   // may overlap
        op += mlen;
    }
    return op;
}

// ---- c-blosc 1.x chunk -----------------------------------------------------------------------
void blosc_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t nbytes_out) {
    if (n < 16) throw ParseError("blosc: short chunk");
    const unsigned flags = src[2];
    const size_t typesize = src[3];
    uint32_t nbytes, blocksize, cbytes;
    std::memcpy(&nbytes, src + 4, 4);
    std::memcpy(&blocksize, src + 8, 4);
    std::memcpy(&cbytes, src + 12, 4);
    if (nbytes != nbytes_out) throw ParseError("blosc: unexpected uncompressed size");
    if (cbytes > n) throw ParseError("blosc: chunk larger than its frame");
    if (typesize == 0) throw ParseError("blosc: zero type size");
    if (flags & 0x2) {                                        // BLOSC_MEMCPYED
        if (n < 16 + (size_t)nbytes) throw ParseError("blosc: truncated memcpy chunk");
        std::memcpy(dst, src + 16, nbytes);
        return;
    }
    if (flags & 0x4) throw ParseError("blosc: bit-shuffle not supported");
    const bool shuffle = (flags & 0x1) && typesize > 1;
    const bool dont_split = (flags & 0x10) != 0;
    const unsigned codec = flags >> 5;                       // 0 blosclz, 1 lz4, 2 snappy, 3 zlib, 4 zstd
    if (codec != 1 && codec != 3) throw ParseError("blosc: only the lz4 and zlib codecs are supported");
    if (blocksize == 0) throw ParseError("blosc: zero block size");
    const size_t nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > n) throw ParseError("blosc: truncated block table");
    std::vector<uint8_t> tmp(blocksize);
    for (size_t blk = 0; blk < nblocks; ++blk) {
        int32_t bstart;
        std::memcpy(&bstart, src + 16 + 4 * blk, 4);
        if (bstart < 0 || (size_t)bstart < 16 + 4 * nblocks || (size_t)bstart >= n) throw ParseError("blosc: block offset outside the chunk");
        if (blk * (size_t)blocksize >= nbytes_out) throw ParseError("blosc: block table larger than the output");
        const size_t bsize = (blk == nblocks - 1 && nbytes % blocksize) ? nbytes % blocksize : blocksize;
        const bool leftover = bsize != blocksize;
        size_t nsplits = 1;
        if (!dont_split && typesize <= 16 && blocksize / typesize >= 128 && !leftover) nsplits = typesize;
        const size_t neblock = bsize / nsplits;
        size_t ip = (size_t)bstart, op = 0;
        uint8_t* target = shuffle ? tmp.data() : dst + blk * blocksize;
        for (size_t s = 0; s < nsplits; ++s) {
            if (ip > n || n - ip < 4) throw ParseError("blosc: truncated split header");
            int32_t cb;
            std::memcpy(&cb, src + ip, 4);
            ip += 4;
            if (cb < 0 || (size_t)cb > n - ip || op + neblock > bsize) throw ParseError("blosc: truncated split");
            if ((size_t)cb == neblock) {
                std::memcpy(target + op, src + ip, neblock);
            } else if (codec == 1) {
                if (lz4_decode(src + ip, (size_t)cb, target + op, neblock) != neblock) throw ParseError("blosc: lz4 size mismatch");
            } else {
                uLongf dl = (uLongf)neblock;
                if (uncompress(target + op, &dl, src + ip, (uLong)cb) != Z_OK || dl != neblock) throw ParseError("blosc: zlib error");
            }
            ip += (size_t)cb;
            op += neblock;
        }
        if (shuffle) {
            uint8_t* out = dst + blk * blocksize;
            const size_t ne = bsize / typesize;
            for (size_t j = 0; j < typesize; ++j)
                for (size_t i = 0; i < ne; ++i) out[i * typesize + j] = tmp[j * ne + i];
            const size_t rem = bsize - ne * typesize;
            std::memcpy(out + ne * typesize, tmp.data() + ne * typesize, rem);
        }
    }
}

// ---- OpenVDB grid ------------------------------------------------------------------------------
enum { COMPRESS_ZIP = 1, COMPRESS_ACTIVE_MASK = 2, COMPRESS_BLOSC = 4 };

struct Tile {
    int32_t org[3];
    int log2dim;
    float val[3];
    bool active;
};
struct Leaf {
    int32_t org[3];
    uint8_t mask[64];
    std::vector<float> vals;        // 512 * ncomp, voxel n -> (x,y,z) = (n>>6, (n>>3)&7, n&7)
};
struct Grid {
    std::string name, type, map_type;
    int ncomp = 1;
    float background[3] = {0, 0, 0};
    double matrix[4][4];            // OpenVDB Mat4d, row-vector convention
    double voxel_size = 1.0;
    std::vector<Tile> tiles;
    std::vector<Leaf> leaves;
    long long active_voxels = 0, active_tiles = 0;
    // dense copy over the active bbox
    int32_t bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
    int dim[3] = {0, 0, 0};
    std::vector<float> dense;       // ncomp == 3 -> float4 (w = 1)
};

inline bool bit(const uint8_t* m, size_t i) { return (m[i >> 3] >> (i & 7)) & 1; }
inline size_t popcount_bytes(const uint8_t* m, size_t nbytes) {
    size_t c = 0;
    for (size_t i = 0; i < nbytes; ++i) c += (size_t)__builtin_popcount(m[i]);
    return c;
}

// the framed (possibly compressed) byte block of io::readData (io/Compression.h)
void read_raw(Reader& r, uint8_t* out, size_t nbytes_out, uint32_t flags) {
    if (flags & (COMPRESS_BLOSC | COMPRESS_ZIP)) {
        const int64_t n = r.get<int64_t>();
        if (n <= 0) {                                        // stored uncompressed
            if ((size_t)(-n) != nbytes_out) throw ParseError("vdb: raw block size mismatch");
            std::memcpy(out, r.take(nbytes_out), nbytes_out);
            return;
        }
        const uint8_t* src = r.take((size_t)n);
        if (flags & COMPRESS_BLOSC) {
            blosc_decode(src, (size_t)n, out, nbytes_out);
        } else {
            uLongf dl = (uLongf)nbytes_out;
            if (uncompress(out, &dl, src, (uLong)n) != Z_OK || dl != nbytes_out) throw ParseError("vdb: zlib error");
        }
        return;
    }
    std::memcpy(out, r.take(nbytes_out), nbytes_out);
}

float half_to_float(uint16_t h);       // defined with the OpenEXR reader below (exact: every binary16 value is a binary32 value)

// the value block of io::readCompressedValues: `count` values of `ncomp` floats.  Grids saved as half float
// (GridDescriptor type suffix "_HalfFloat", io/GridDescriptor.cc; Houdini's "16-bit float" save option) store this block
// -- and only this block -- as binary16 (io::HalfReader): 2 bytes per component before compression
void read_value_block(Reader& r, float* out, size_t count, int ncomp, uint32_t flags, bool half) {
    if (!half) {
        read_raw(r, (uint8_t*)out, count * ncomp * 4, flags);
        return;
    }
    std::vector<uint16_t> h(count * ncomp);
    read_raw(r, (uint8_t*)h.data(), count * ncomp * 2, flags);
    for (size_t i = 0; i < count * (size_t)ncomp; ++i) out[i] = half_to_float(h[i]);
}

// io::readCompressedValues: `n` values of `ncomp` floats, value mask `vm` (n bits).  The inactive values of the
// node-mask compression are full floats even in a half-float grid (the writer truncates them to half precision but
// stores 4 bytes, io/Compression.h)
void read_values(Reader& r, size_t n, const uint8_t* vm, uint32_t flags, int ncomp, const float* background, std::vector<float>& out, bool half) {
    const int8_t metadata = r.get<int8_t>();
    float inactive0[3], inactive1[3];
    for (int c = 0; c < ncomp; ++c) inactive0[c] = inactive1[c] = background[c];
    if (metadata == 1)
        for (int c = 0; c < ncomp; ++c) inactive0[c] = -background[c];
    if (metadata == 2 || metadata == 4 || metadata == 5) std::memcpy(inactive0, r.take(4 * ncomp), 4 * ncomp);
    if (metadata == 5) std::memcpy(inactive1, r.take(4 * ncomp), 4 * ncomp);
    const uint8_t* sel = nullptr;
    if (metadata == 3 || metadata == 4 || metadata == 5) sel = r.take(n / 8);
    const size_t count = ((flags & COMPRESS_ACTIVE_MASK) && metadata != 6) ? popcount_bytes(vm, n / 8) : n;
    out.assign(n * ncomp, 0.0f);
    if (count == n) {
        read_value_block(r, out.data(), n, ncomp, flags, half);
        return;
    }
    std::vector<float> packed(count * ncomp);
    read_value_block(r, packed.data(), count, ncomp, flags, half);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        float* o = &out[i * ncomp];
        if (bit(vm, i)) {
            for (int c = 0; c < ncomp; ++c) o[c] = packed[k * ncomp + c];
            ++k;
        } else {
            const float* v = (sel && bit(sel, i)) ? inactive1 : inactive0;
            for (int c = 0; c < ncomp; ++c) o[c] = v[c];
        }
    }
}

void skip_meta_value(Reader& r) {
    const uint32_t size = r.get<uint32_t>();
    r.take(size);
}

void read_transform(Reader& r, Grid& g) {
    g.map_type = r.str();
    double (*m)[4] = g.matrix;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m[i][j] = i == j ? 1.0 : 0.0;
    auto f64s = [&](double* v, int k) { std::memcpy(v, r.take(8 * (size_t)k), 8 * (size_t)k); };
    if (g.map_type == "UniformScaleMap" || g.map_type == "ScaleMap") {
        double v[15];
        f64s(v, 15);
        m[0][0] = v[0]; m[1][1] = v[1]; m[2][2] = v[2];
        g.voxel_size = v[3];
    } else if (g.map_type == "UniformScaleTranslateMap" || g.map_type == "ScaleTranslateMap") {
        double v[18];
        f64s(v, 18);
        m[0][0] = v[3]; m[1][1] = v[4]; m[2][2] = v[5];
        m[3][0] = v[0]; m[3][1] = v[1]; m[3][2] = v[2];
        g.voxel_size = v[6];
    } else if (g.map_type == "TranslationMap") {
        double v[3];
        f64s(v, 3);
        m[3][0] = v[0]; m[3][1] = v[1]; m[3][2] = v[2];
    } else if (g.map_type == "AffineMap") {
        double v[16];
        f64s(v, 16);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) m[i][j] = v[i * 4 + j];
        g.voxel_size = std::sqrt(m[0][0] * m[0][0] + m[0][1] * m[0][1] + m[0][2] * m[0][2]);
    } else {
        throw ParseError("vdb: unsupported transform map " + g.map_type);
    }
}

void densify(Grid& g) {
    // evalActiveVoxelBoundingBox: active voxels of leaves + active tiles
    int64_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (const Leaf& lf : g.leaves)
        for (int n = 0; n < 512; ++n)
            if (bit(lf.mask, (size_t)n)) {
                const int64_t p[3] = {lf.org[0] + (n >> 6), lf.org[1] + ((n >> 3) & 7), lf.org[2] + (n & 7)};
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
                g.active_voxels++;
            }
    for (const Tile& t : g.tiles)
        if (t.active) {
            const int64_t s = (int64_t)1 << t.log2dim;
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (int64_t)t.org[a]); hi[a] = std::max(hi[a], (int64_t)t.org[a] + s - 1); }
            g.active_tiles++;
            g.active_voxels += s * s * s;
        }
    if (lo[0] > hi[0]) throw ParseError("vdb: grid '" + g.name + "' has no active voxels");
    for (int a = 0; a < 3; ++a) {
        g.bbox_min[a] = (int32_t)lo[a];
        g.bbox_max[a] = (int32_t)hi[a];
        g.dim[a] = (int)(hi[a] - lo[a] + 1);
    }
    const size_t nvox = (size_t)g.dim[0] * g.dim[1] * g.dim[2];
    if (nvox > ((size_t)1 << 36)) throw ParseError("vdb: dense bounding box too large");
    const int oc = g.ncomp == 3 ? 4 : 1;                     // Vec3s -> float4 with w = 1 (gpu_vdb.cpp:67-72)
    g.dense.assign(nvox * oc, 0.0f);
    auto put = [&](int64_t x, int64_t y, int64_t z, const float* v) {
        if (x < lo[0] || x > hi[0] || y < lo[1] || y > hi[1] || z < lo[2] || z > hi[2]) return;
        const size_t idx = ((size_t)(z - lo[2]) * g.dim[1] + (size_t)(y - lo[1])) * g.dim[0] + (size_t)(x - lo[0]);   // LayoutXYZ
        if (oc == 1) g.dense[idx] = v[0];
        else { g.dense[4 * idx] = v[0]; g.dense[4 * idx + 1] = v[1]; g.dense[4 * idx + 2] = v[2]; g.dense[4 * idx + 3] = 1.0f; }
    };
    // background everywhere, then tiles (copyToDense copies tile values active or not), then leaves
    for (size_t i = 0; i < nvox; ++i) {
        if (oc == 1) g.dense[i] = g.background[0];
        else { g.dense[4 * i] = g.background[0]; g.dense[4 * i + 1] = g.background[1]; g.dense[4 * i + 2] = g.background[2]; g.dense[4 * i + 3] = 1.0f; }
    }
    for (const Tile& t : g.tiles) {
        const int64_t s = (int64_t)1 << t.log2dim;
        const int64_t x0 = std::max<int64_t>(t.org[0], lo[0]), x1 = std::min<int64_t>(t.org[0] + s - 1, hi[0]);
        const int64_t y0 = std::max<int64_t>(t.org[1], lo[1]), y1 = std::min<int64_t>(t.org[1] + s - 1, hi[1]);
        const int64_t z0 = std::max<int64_t>(t.org[2], lo[2]), z1 = std::min<int64_t>(t.org[2] + s - 1, hi[2]);
        for (int64_t z = z0; z <= z1; ++z)
            for (int64_t y = y0; y <= y1; ++y)
                for (int64_t x = x0; x <= x1; ++x) put(x, y, z, t.val);
    }
    for (const Leaf& lf : g.leaves)
        for (int n = 0; n < 512; ++n) put(lf.org[0] + (n >> 6), lf.org[1] + ((n >> 3) & 7), lf.org[2] + (n & 7), &lf.vals[(size_t)n * g.ncomp]);
}

void read_grid(Reader& r, Grid& g, int64_t grid_pos, int64_t block_pos, int64_t end_pos) {
    if (g.type.rfind("Tree_float_5_4_3", 0) == 0) g.ncomp = 1;
    else if (g.type.rfind("Tree_vec3s_5_4_3", 0) == 0) g.ncomp = 3;
    else throw ParseError("vdb: unsupported grid type " + g.type);
    // the descriptor's type suffix is what OpenVDB's reader goes by (GridDescriptor::read strips it and sets saveFloatAsHalf,
    // Archive::readGrid passes that to the stream); the grid metadatum of the same name is informational
    const bool half = g.type.size() >= 10 && g.type.compare(g.type.size() - 10, 10, "_HalfFloat") == 0;
    r.seek(grid_pos);
    const uint32_t flags = r.get<uint32_t>();
    const uint32_t meta_count = r.get<uint32_t>();
    for (uint32_t i = 0; i < meta_count; ++i) {
        const std::string mname = r.str();
        const std::string mtype = r.str();
        if (mname == "is_saved_as_half_float" && mtype == "bool") {
            // OpenVDB writes this metadatum and the descriptor suffix from the same flag (GridBase::saveFloatAsHalf); a file where
            // they disagree would be decoded with the wrong value width and fail far away ("did not parse to its end offset")
            const uint32_t size = r.get<uint32_t>();
            const uint8_t* v = r.take(size);
            const bool meta_half = size >= 1 && v[0] != 0;
            if (meta_half != half)
                throw ParseError("vdb: grid '" + g.name + "': descriptor type " + g.type + (half ? " says" : " does not say") +
                                 " half-float values but its is_saved_as_half_float metadatum says " + (meta_half ? "true" : "false"));
            continue;
        }
        skip_meta_value(r);
    }
    read_transform(r, g);
    // ---- topology: RootNode<InternalNode<InternalNode<LeafNode<T,3>,4>,5>>
    const int32_t buffer_count = r.get<int32_t>();
    if (buffer_count != 1) throw ParseError("vdb: multi-buffer trees are not supported");
    std::memcpy(g.background, r.take(4 * (size_t)g.ncomp), 4 * (size_t)g.ncomp);
    const uint32_t num_tiles = r.get<uint32_t>();
    const uint32_t num_children = r.get<uint32_t>();
    for (uint32_t i = 0; i < num_tiles; ++i) {
        Tile t;
        std::memcpy(t.org, r.take(12), 12);
        std::memset(t.val, 0, sizeof(t.val));
        std::memcpy(t.val, r.take(4 * (size_t)g.ncomp), 4 * (size_t)g.ncomp);
        t.active = r.get<uint8_t>() != 0;
        t.log2dim = 12;
        g.tiles.push_back(t);
    }
    std::vector<std::vector<int32_t>> leaf_origins;
    std::vector<float> vals;
    auto grab = [&](size_t k) {
        const uint8_t* s = r.take(k);
        return std::vector<uint8_t>(s, s + k);
    };
    auto differs = [&](const float* v) {
        for (int c = 0; c < g.ncomp; ++c)
            if (v[c] != g.background[c]) return true;
        return false;
    };
    for (uint32_t ch = 0; ch < num_children; ++ch) {
        int32_t org5[3];
        std::memcpy(org5, r.take(12), 12);
        const std::vector<uint8_t> cm5 = grab(4096), vm5 = grab(4096);
        read_values(r, 32768, vm5.data(), flags, g.ncomp, g.background, vals, half);
        std::vector<float> vals5 = vals;
        for (size_t i = 0; i < 32768; ++i) {
            const int32_t o5[3] = {org5[0] + (int32_t)((i >> 10) << 7), org5[1] + (int32_t)(((i >> 5) & 31) << 7), org5[2] + (int32_t)((i & 31) << 7)};
            if (!bit(cm5.data(), i)) {
                const bool act = bit(vm5.data(), i);
                if (act || differs(&vals5[i * g.ncomp])) {
                    Tile t;
                    std::memcpy(t.org, o5, 12);
                    std::memset(t.val, 0, sizeof(t.val));
                    for (int c = 0; c < g.ncomp; ++c) t.val[c] = vals5[i * g.ncomp + c];
                    t.active = act;
                    t.log2dim = 7;
                    g.tiles.push_back(t);
                }
                continue;
            }
            const std::vector<uint8_t> cm4 = grab(512), vm4 = grab(512);
            read_values(r, 4096, vm4.data(), flags, g.ncomp, g.background, vals, half);
            for (size_t j = 0; j < 4096; ++j) {
                const int32_t o4[3] = {o5[0] + (int32_t)((j >> 8) << 3), o5[1] + (int32_t)(((j >> 4) & 15) << 3), o5[2] + (int32_t)((j & 15) << 3)};
                if (!bit(cm4.data(), j)) {
                    const bool act = bit(vm4.data(), j);
                    if (act || differs(&vals[j * g.ncomp])) {
                        Tile t;
                        std::memcpy(t.org, o4, 12);
                        std::memset(t.val, 0, sizeof(t.val));
                        for (int c = 0; c < g.ncomp; ++c) t.val[c] = vals[j * g.ncomp + c];
                        t.active = act;
                        t.log2dim = 3;
                        g.tiles.push_back(t);
                    }
                    continue;
                }
                r.take(64);                                   // leaf value mask (topology pass)
                leaf_origins.push_back({o4[0], o4[1], o4[2]});
            }
        }
    }
    // ---- buffers, same depth-first order
    r.seek(block_pos);
    g.leaves.resize(leaf_origins.size());
    for (size_t i = 0; i < leaf_origins.size(); ++i) {
        Leaf& lf = g.leaves[i];
        std::memcpy(lf.org, leaf_origins[i].data(), 12);
        std::memcpy(lf.mask, r.take(64), 64);
        read_values(r, 512, lf.mask, flags, g.ncomp, g.background, lf.vals, half);
    }
    if ((int64_t)r.p != end_pos) throw ParseError("vdb: grid '" + g.name + "' did not parse to its end offset");
    densify(g);
}

}  // namespace

struct vpt_io_volume {
    Grid grids[3];          // density, emission, colour
    bool present[3] = {false, false, false};
    vpt_gpu_vdb info;
};

struct vpt_io_ins {
    bool light_file = false;
    std::vector<std::string> files;
    std::vector<std::vector<vpt_io_instance>> instances;
    std::vector<vpt_point_light> lights;
};

extern "C" {

const char* vpt_io_last_error(void) { return g_io_error.c_str(); }

// test probe (include/vpt_testhooks.h): one c-blosc chunk through the decoder; 0 on success, VPT_E_IO on a ParseError
int vpt_io_test_blosc_decode(const unsigned char* src, size_t n, unsigned char* dst, size_t nbytes_out) {
    try {
        blosc_decode(src, n, dst, nbytes_out);
    } catch (const std::exception& e) {
        return fail(VPT_E_IO, "%s", e.what());
    }
    return VPT_OK;
}
void vpt_io_free(void* p) { free(p); }

int vpt_io_vdb_load(const char* filename, const char* density_channel, const char* emission_channel, const char* color_channel,
                    vpt_io_volume** out) {
    if (!filename || !out || !density_channel || !density_channel[0]) return fail(VPT_E_INVALID, "vpt_io_vdb_load: file name / density channel can't be empty");
    *out = nullptr;
    std::vector<uint8_t> buf;
    if (!read_file(filename, buf)) return fail(VPT_E_IO, "File doesn't exist or is unreadable: %s", filename);
    std::unique_ptr<vpt_io_volume> vol(new vpt_io_volume());
    const std::string want[3] = {density_channel, emission_channel ? emission_channel : "", color_channel ? color_channel : ""};
    try {
        Reader r{buf.data(), buf.size(), 0};
        if (r.get<int64_t>() != 0x56444220) throw ParseError("not an OpenVDB file");
        const uint32_t version = r.get<uint32_t>();
        if (version < 222) throw ParseError("vdb: file format version < 222 is not supported");
        r.get<uint32_t>();
        r.get<uint32_t>();
        const uint8_t has_offsets = r.get<uint8_t>();
        if (!has_offsets) throw ParseError("vdb: files without grid offsets are not supported");
        r.take(36);                                           // uuid
        const uint32_t nmeta = r.get<uint32_t>();
        for (uint32_t i = 0; i < nmeta; ++i) {
            r.str();
            r.str();
            skip_meta_value(r);
        }
        const uint32_t ngrids = r.get<uint32_t>();
        for (uint32_t gi = 0; gi < ngrids; ++gi) {
            std::string name = r.str();
            const std::string type = r.str();
            r.str();                                          // instance parent
            const int64_t grid_pos = r.get<int64_t>(), block_pos = r.get<int64_t>(), end_pos = r.get<int64_t>();
            // unique-name suffix of multi-grid files: "name\x1e<n>"
            const size_t sep = name.find('\x1e');
            if (sep != std::string::npos) name.resize(sep);
            for (int w = 0; w < 3; ++w) {
                // the reference's if / else-if chain: a grid is taken by the first channel that names it
                if (want[w].empty() || name != want[w]) continue;
                bool earlier = false;
                for (int e = 0; e < w; ++e) earlier |= (!want[e].empty() && name == want[e]);
                if (earlier) continue;
                Grid& g = vol->grids[w];
                g = Grid();
                g.name = name;
                g.type = type;
                Reader rg{buf.data(), buf.size(), 0};
                read_grid(rg, g, grid_pos, block_pos, end_pos);
                if ((w < 2 && g.ncomp != 1) || (w == 2 && g.ncomp != 3)) throw ParseError("vdb: grid '" + name + "' has the wrong value type for its channel");
                vol->present[w] = true;
            }
            r.seek(end_pos);
        }
    } catch (const std::exception& e) {
        return fail(VPT_E_IO, "%s: %s", filename, e.what());
    }
    if (!vol->present[0]) return fail(VPT_E_IO, "%s: no float grid named '%s'", filename, density_channel);
    // VDB_INFO, gpu_vdb.cpp:123-126, 199-212, 453-470
    const Grid& d = vol->grids[0];
    vpt_gpu_vdb& v = vol->info;
    std::memset(&v, 0, sizeof(v));
    float mx = .0f, mn = 3.402823466e+38f;
    for (float val : d.dense) {
        mx = fmaxf(mx, val);
        mn = fminf(fmaxf(1.192092896e-07f, val), mn);
    }
    v.vdb_info.max_density = mx;
    v.vdb_info.min_density = mn;
    v.vdb_info.has_emission = vol->present[1];
    v.vdb_info.has_color = vol->present[2];
    v.vdb_info.bmin = {(float)d.bbox_min[0], (float)d.bbox_min[1], (float)d.bbox_min[2]};
    v.vdb_info.bmax = {(float)d.bbox_max[0], (float)d.bbox_max[1], (float)d.bbox_max[2]};
    v.vdb_info.dim = {d.dim[0], d.dim[1], d.dim[2]};
    v.vdb_info.voxelsize = (float)d.voxel_size;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) v.xform[i][j] = (float)d.matrix[j][i];                  // convert_to_mat4, gpu_vdb.cpp:81-92
    *out = vol.release();
    return VPT_OK;
}

void vpt_io_vdb_free(vpt_io_volume* vol) { delete vol; }

int vpt_io_vdb_info(const vpt_io_volume* vol, vpt_gpu_vdb* out) {
    if (!vol || !out) return VPT_E_INVALID;
    *out = vol->info;
    return VPT_OK;
}

int vpt_io_vdb_grid(const vpt_io_volume* vol, int which, const float** data, vpt_int3* dim) {
    if (!vol || which < 0 || which > 2) return VPT_E_INVALID;
    if (!vol->present[which]) return fail(VPT_E_NOT_READY, "vpt_io_vdb_grid: channel %d was not loaded", which);
    const Grid& g = vol->grids[which];
    if (data) *data = g.dense.data();
    if (dim) *dim = {g.dim[0], g.dim[1], g.dim[2]};
    return VPT_OK;
}

int vpt_io_vdb_stats(const vpt_io_volume* vol, int which, long long out[3]) {
    if (!vol || which < 0 || which > 2 || !out) return VPT_E_INVALID;
    if (!vol->present[which]) return VPT_E_NOT_READY;
    const Grid& g = vol->grids[which];
    out[0] = (long long)g.leaves.size();
    out[1] = g.active_voxels;
    out[2] = g.active_tiles;
    return VPT_OK;
}

int vpt_io_vdb_upload(vpt_ctx* ctx, const vpt_io_volume* vol, vpt_gpu_vdb* out) {
    if (!ctx || !vol || !out) return VPT_E_INVALID;
    *out = vol->info;
    vpt_texture_t* handle[3] = {&out->vdb_info.density_texture, &out->vdb_info.emission_texture, &out->vdb_info.color_texture};
    for (int w = 0; w < 3; ++w) {
        if (!vol->present[w]) continue;
        const Grid& g = vol->grids[w];
        // sampler state of gpu_vdb.cpp:235-248, 314-327, 394-407: normalised, linear, clamp
        vpt_texture_desc d = {g.dim[0], g.dim[1], g.dim[2], w == 2 ? 4 : 1, 1, VPT_FILTER_LINEAR, {VPT_ADDR_CLAMP, VPT_ADDR_CLAMP, VPT_ADDR_CLAMP}};
        const int rc = vpt_texture_create(ctx, &d, g.dense.data(), handle[w]);
        if (rc != VPT_OK) return rc;
    }
    return VPT_OK;
}

// ---- .ins (main.cpp:980-1056) -------------------------------------------------------------------
int vpt_io_ins_read(const char* filename, vpt_io_ins** out) {
    if (!filename || !filename[0] || !out) return VPT_E_INVALID;
    *out = nullptr;
    std::ifstream stream(filename);
    if (!stream) return fail(VPT_E_IO, "cannot open instance file %s", filename);
    std::unique_ptr<vpt_io_ins> ins(new vpt_io_ins());
    auto chomp = [](std::string& s) {
        while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
    };
    std::string first;
    std::getline(stream, first);
    chomp(first);
    if (first == "light") {
        ins->light_file = true;
        std::string line;
        std::getline(stream, line);
        int num_lights = 0;
        std::istringstream(line) >> num_lights;
        if (num_lights < 0) return fail(VPT_E_IO, "%s: negative light count", filename);
        for (int i = 0; i < num_lights; ++i) {
            std::getline(stream, line);
            std::istringstream params(line);
            double px = 0, py = 0, pz = 0, cr = 0, cg = 0, cb = 0, p = 0;
            params >> px >> py >> pz >> cr >> cg >> cb >> p;
            vpt_point_light l;
            std::memset(&l, 0, sizeof(l));                  // point_light(): pos 0, color/power set below
            l.color = {(float)cr, (float)cg, (float)cb};
            l.pos = {(float)px, (float)py, (float)pz};
            l.power = (float)p;
            ins->lights.push_back(l);
        }
    } else {
        int num_volumes = 0;
        std::istringstream(first) >> num_volumes;
        if (num_volumes < 0) return fail(VPT_E_IO, "%s: negative file count", filename);
        for (int i = 0; i < num_volumes; ++i) {
            std::string name, line;
            std::getline(stream, name);
            chomp(name);
            std::getline(stream, line);
            unsigned n = 0;
            std::istringstream(line) >> n;
            std::vector<vpt_io_instance> v(n);
            for (unsigned x = 0; x < n; ++x) {
                std::getline(stream, line);
                std::istringstream params(line);
                double p1 = 0, p2 = 0, p3 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, s = 0;
                params >> p1 >> p2 >> p3 >> r1 >> r2 >> r3 >> r4 >> s;
                v[x].position[0] = p1; v[x].position[1] = p2; v[x].position[2] = p3;
                v[x].rotation[0] = r1; v[x].rotation[1] = r2; v[x].rotation[2] = r3; v[x].rotation[3] = r4;
                v[x].scale = s;
            }
            ins->files.push_back(name);
            ins->instances.push_back(std::move(v));
        }
    }
    *out = ins.release();
    return VPT_OK;
}
void vpt_io_ins_free(vpt_io_ins* ins) { delete ins; }
int vpt_io_ins_is_light_file(const vpt_io_ins* ins) { return ins && ins->light_file ? 1 : 0; }
int vpt_io_ins_num_files(const vpt_io_ins* ins) { return ins ? (int)ins->files.size() : 0; }
const char* vpt_io_ins_file_name(const vpt_io_ins* ins, int file) {
    return (ins && file >= 0 && file < (int)ins->files.size()) ? ins->files[(size_t)file].c_str() : nullptr;
}
int vpt_io_ins_num_instances(const vpt_io_ins* ins, int file) {
    return (ins && file >= 0 && file < (int)ins->files.size()) ? (int)ins->instances[(size_t)file].size() : 0;
}
const vpt_io_instance* vpt_io_ins_instances(const vpt_io_ins* ins, int file) {
    return (ins && file >= 0 && file < (int)ins->files.size()) ? ins->instances[(size_t)file].data() : nullptr;
}
int vpt_io_ins_num_lights(const vpt_io_ins* ins) { return ins ? (int)ins->lights.size() : 0; }
const vpt_point_light* vpt_io_ins_lights(const vpt_io_ins* ins) { return ins ? ins->lights.data() : nullptr; }

// ---- BMP (24-bit, uncompressed) -------------------------------------------------------------------
int vpt_io_load_bmp(const char* filename, float** rgb, int* width, int* height) {
    if (!filename || !rgb || !width || !height) return VPT_E_INVALID;
    std::vector<uint8_t> b;
    if (!read_file(filename, b)) return fail(VPT_E_IO, "Unable to load file %s", filename);
    if (b.size() < 54 || b[0] != 'B' || b[1] != 'M') return fail(VPT_E_IO, "%s: not a BMP file", filename);
    uint32_t off;
    int32_t w, h;
    uint16_t bpp;
    uint32_t comp;
    std::memcpy(&off, &b[10], 4);
    std::memcpy(&w, &b[18], 4);
    std::memcpy(&h, &b[22], 4);
    std::memcpy(&bpp, &b[28], 2);
    std::memcpy(&comp, &b[30], 4);
    if (bpp != 24 || comp != 0 || w <= 0 || h == 0) return fail(VPT_E_IO, "%s: only uncompressed 24-bit BMP is supported", filename);
    const bool bottom_up = h > 0;
    const int H = h > 0 ? h : -h;
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    if (b.size() < off + stride * (size_t)H) return fail(VPT_E_IO, "%s: truncated BMP", filename);
    float* out = (float*)malloc(sizeof(float) * 3 * (size_t)w * H);
    if (!out) return VPT_E_NOMEM;
    for (int y = 0; y < H; ++y) {
        const uint8_t* row = &b[off + stride * (size_t)(bottom_up ? H - 1 - y : y)];
        for (int x = 0; x < w; ++x) {
            const float bl = row[3 * x], gr = row[3 * x + 1], rd = row[3 * x + 2];
            float* o = out + 3 * ((size_t)y * w + x);
            o[0] = rd / 255.0f;           // fileIO.cpp:482-484: .x = red, .y = BLUE, .z = GREEN
            o[1] = bl / 255.0f;
            o[2] = gr / 255.0f;
        }
    }
    *rgb = out;
    *width = w;
    *height = H;
    return VPT_OK;
}

// ---- OpenEXR scanline (NO / ZIPS / ZIP compression, HALF / FLOAT channels) -----------------------
namespace {
float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024)) { mm <<= 1; ++sh; }
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 1023) << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
struct ExrChannel {
    std::string name;
    int type;       // 0 uint, 1 half, 2 float
};
}  // namespace

int vpt_io_load_exr_rgb(const char* filename, float** rgb, int* width, int* height) {
    if (!filename || !rgb || !width || !height) return VPT_E_INVALID;
    std::vector<uint8_t> b;
    if (!read_file(filename, b)) return fail(VPT_E_IO, "Unable to load file %s", filename);
    try {
        Reader r{b.data(), b.size(), 0};
        if (r.get<uint32_t>() != 20000630u) throw ParseError("not an OpenEXR file");
        const uint32_t ver = r.get<uint32_t>();
        if (ver & 0x1E00u) throw ParseError("exr: tiled / deep / multi-part files are not supported");
        std::vector<ExrChannel> chans;
        int compression = -1;
        int32_t dw[4] = {0, 0, -1, -1};
        auto cstr = [&]() {
            std::string s;
            for (;;) {
                const char c = (char)r.get<uint8_t>();
                if (!c) break;
                s.push_back(c);
            }
            return s;
        };
        for (;;) {
            const std::string name = cstr();
            if (name.empty()) break;
            const std::string type = cstr();
            const uint32_t size = r.get<uint32_t>();
            const size_t end = r.p + size;
            if (name == "channels") {
                for (;;) {
                    const std::string cn = cstr();
                    if (cn.empty()) break;
                    ExrChannel c;
                    c.name = cn;
                    c.type = r.get<int32_t>();
                    r.take(4);                                // pLinear + reserved
                    const int32_t xs = r.get<int32_t>(), ys = r.get<int32_t>();
                    if (xs != 1 || ys != 1) throw ParseError("exr: subsampled channels are not supported");
                    chans.push_back(c);
                }
            } else if (name == "compression") {
                compression = r.get<uint8_t>();
            } else if (name == "dataWindow") {
                std::memcpy(dw, r.take(16), 16);
            }
            r.p = end;
            if (r.p > r.n) throw ParseError("exr: truncated header");
        }
        const int W = dw[2] - dw[0] + 1, H = dw[3] - dw[1] + 1;
        if (W <= 0 || H <= 0 || chans.empty()) throw ParseError("exr: bad data window / no channels");
        if (compression != 0 && compression != 2 && compression != 3) throw ParseError("exr: only NONE / ZIPS / ZIP compression is supported");
        const int lines_per_chunk = compression == 3 ? 16 : 1;
        const int nchunks = (H + lines_per_chunk - 1) / lines_per_chunk;
        std::vector<uint64_t> offsets((size_t)nchunks);
        std::memcpy(offsets.data(), r.take(8 * (size_t)nchunks), 8 * (size_t)nchunks);
        size_t line_bytes = 0;
        for (const ExrChannel& c : chans) line_bytes += (size_t)W * (c.type == 1 ? 2 : 4);
        int ci[3] = {-1, -1, -1};
        for (size_t i = 0; i < chans.size(); ++i) {
            if (chans[i].name == "R") ci[0] = (int)i;
            if (chans[i].name == "G") ci[1] = (int)i;
            if (chans[i].name == "B") ci[2] = (int)i;
        }
        if (ci[0] < 0 && chans.size() == 1) ci[0] = ci[1] = ci[2] = 0;       // luminance-only file
        if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) throw ParseError("exr: R, G, B channels not found");
        float* out = (float*)malloc(sizeof(float) * 3 * (size_t)W * H);
        if (!out) return VPT_E_NOMEM;
        std::vector<uint8_t> raw, tmp;
        for (int ch = 0; ch < nchunks; ++ch) {
            Reader c{b.data(), b.size(), (size_t)offsets[(size_t)ch]};
            const int32_t y0 = c.get<int32_t>();
            const uint32_t size = c.get<uint32_t>();
            const uint8_t* data = c.take(size);
            const int nlines = std::min(lines_per_chunk, dw[3] - y0 + 1);
            const size_t want = line_bytes * (size_t)nlines;
            raw.resize(want);
            if (compression == 0 || size == want) {
                std::memcpy(raw.data(), data, want);
            } else {
                tmp.resize(want);
                uLongf dl = (uLongf)want;
                if (uncompress(tmp.data(), &dl, data, size) != Z_OK || dl != want) { free(out); throw ParseError("exr: zlib error"); }
                for (size_t i = 1; i < want; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);     // predictor
                const size_t half = (want + 1) / 2;                                                  // de-interleave
                for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
            }
            for (int l = 0; l < nlines; ++l) {
                const int y = y0 - dw[1] + l;
                const uint8_t* p = raw.data() + line_bytes * (size_t)l;
                for (size_t k = 0; k < chans.size(); ++k) {       // channels are stored sorted by name, one plane per line
                    const size_t bytes = chans[k].type == 1 ? 2 : 4;
                    for (int comp = 0; comp < 3; ++comp) {
                        if (ci[comp] != (int)k) continue;
                        for (int x = 0; x < W; ++x) {
                            float v;
                            if (chans[k].type == 1) { uint16_t hv; std::memcpy(&hv, p + 2 * (size_t)x, 2); v = half_to_float(hv); }
                            else if (chans[k].type == 2) std::memcpy(&v, p + 4 * (size_t)x, 4);
                            else { uint32_t uv; std::memcpy(&uv, p + 4 * (size_t)x, 4); v = (float)uv; }
                            out[3 * ((size_t)y * W + x) + comp] = v;
                        }
                    }
                    p += bytes * (size_t)W;
                }
            }
        }
        *rgb = out;
        *width = W;
        *height = H;
    } catch (const std::exception& e) {
        return fail(VPT_E_IO, "%s: %s", filename, e.what());
    }
    return VPT_OK;
}

// ---- Radiance RGBE (hdr_loader.h:60-277) -----------------------------------------------------------
int vpt_io_load_hdr(const char* filename, float** rgba, int* width, int* height) {
    if (!filename || !rgba || !width || !height) return VPT_E_INVALID;
    std::vector<uint8_t> b;
    if (!read_file(filename, b)) return fail(VPT_E_IO, "error loading environment map file %s", filename);
    size_t p = 0;
    auto line = [&]() {
        std::string s;
        while (p < b.size() && b[p] != '\n') s.push_back((char)b[p++]);
        if (p < b.size()) ++p;
        return s;
    };
    std::string l = line();
    if (l.rfind("#?", 0) != 0) return fail(VPT_E_IO, "%s: not a Radiance HDR file", filename);
    for (;;) {
        if (p >= b.size()) return fail(VPT_E_IO, "%s: truncated HDR header", filename);
        l = line();
        if (l.empty()) break;
    }
    l = line();
    int H = 0, W = 0;
    if (sscanf(l.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) return fail(VPT_E_IO, "%s: unsupported HDR orientation '%s'", filename, l.c_str());
    float* out = (float*)calloc((size_t)W * H, sizeof(float) * 4);
    if (!out) return VPT_E_NOMEM;
    std::vector<uint8_t> sl((size_t)W * 4);
    for (int y = 0; y < H; ++y) {
        bool ok = true;
        if (W >= 8 && W < 32768 && p + 4 <= b.size() && b[p] == 2 && b[p + 1] == 2 && !(b[p + 2] & 0x80) && ((b[p + 2] << 8) | b[p + 3]) == W) {
            p += 4;                                          // new-style RLE: four planes
            for (int c = 0; c < 4 && ok; ++c) {
                int x = 0;
                while (x < W && ok) {
                    if (p >= b.size()) { ok = false; break; }
                    int cnt = b[p++];
                    if (cnt > 128) {
                        cnt &= 127;
                        if (p >= b.size() || x + cnt > W) { ok = false; break; }
                        const uint8_t v = b[p++];
                        for (int i = 0; i < cnt; ++i) sl[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (cnt == 0 || p + (size_t)cnt > b.size() || x + cnt > W) { ok = false; break; }
                        for (int i = 0; i < cnt; ++i) sl[(size_t)(x++) * 4 + c] = b[p++];
                    }
                }
            }
        } else {
            if (p + (size_t)W * 4 > b.size()) ok = false;
            else { std::memcpy(sl.data(), &b[p], (size_t)W * 4); p += (size_t)W * 4; }
        }
        if (!ok) { free(out); return fail(VPT_E_IO, "%s: corrupt HDR scanline %d", filename, y); }
        for (int x = 0; x < W; ++x) {
            const uint8_t* e = &sl[(size_t)x * 4];
            float* o = out + 4 * ((size_t)y * W + x);
            if (e[3] != 0) {                                  // hdr_rgbe_to_color, hdr_loader.h:193-212
                uint32_t ui = (uint32_t)(((int)e[3] - 9) << 23) & 0x7f800000u;
                float f;
                std::memcpy(&f, &ui, 4);
                o[0] = (float)(e[0] + 0.5f) * f;
                o[1] = (float)(e[1] + 0.5f) * f;
                o[2] = (float)(e[2] + 0.5f) * f;
            }
        }
    }
    *rgba = out;
    *width = W;
    *height = H;
    return VPT_OK;
}

// ---- writers ------------------------------------------------------------------------------------------
int vpt_io_write_pfm(const char* filename, const float* pixels, int channels, int width, int height) {
    if (!filename || !pixels || (channels != 3 && channels != 4 && channels != 1) || width <= 0 || height <= 0) return VPT_E_INVALID;
    FILE* f = fopen(filename, "wb");
    if (!f) return fail(VPT_E_IO, "cannot write %s", filename);
    fprintf(f, "%s\n%d %d\n-1.0\n", channels == 1 ? "Pf" : "PF", width, height);
    std::vector<float> row((size_t)width * (channels == 1 ? 1 : 3));
    for (int y = height - 1; y >= 0; --y) {
        for (int x = 0; x < width; ++x) {
            const float* s = pixels + (size_t)channels * ((size_t)y * width + x);
            if (channels == 1) row[(size_t)x] = s[0];
            else { row[3 * (size_t)x] = s[0]; row[3 * (size_t)x + 1] = s[1]; row[3 * (size_t)x + 2] = s[2]; }
        }
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return VPT_OK;
}

int vpt_io_write_ppm(const char* filename, const unsigned int* display, int width, int height) {
    if (!filename || !display || width <= 0 || height <= 0) return VPT_E_INVALID;
    FILE* f = fopen(filename, "wb");
    if (!f) return fail(VPT_E_IO, "cannot write %s", filename);
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> row((size_t)width * 3);
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const unsigned int v = display[(size_t)y * width + x];       // 0xffRRGGBB (render_kernel.cu:2311)
            row[3 * (size_t)x] = (uint8_t)(v >> 16);
            row[3 * (size_t)x + 1] = (uint8_t)(v >> 8);
            row[3 * (size_t)x + 2] = (uint8_t)v;
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return VPT_OK;
}

// PNG (ISO/IEC 15948): signature, IHDR, one IDAT holding the zlib stream of the filtered scanlines (filter type 0 per row), IEND; chunk CRCs and
// the deflate stream come from the zlib this library already links for ZIP-compressed VDB / EXR data.  The reference writes its display buffer through
// OpenImageIO as four 8-bit channels (save_texture_png(uint32_t*), fileIO.cpp:140-154); here the 0xffRRGGBB words are unpacked to R, G, B (, A).
namespace {
void png_chunk(FILE* f, const char tag[4], const uint8_t* data, size_t n) {
    uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
    fwrite(len, 1, 4, f);
    fwrite(tag, 1, 4, f);
    if (n) fwrite(data, 1, n, f);
    uLong c = crc32(0L, (const Bytef*)tag, 4);
    if (n) c = crc32(c, (const Bytef*)data, (uInt)n);
    uint8_t crc[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    fwrite(crc, 1, 4, f);
}
int write_png8(const char* filename, const std::vector<uint8_t>& raw, int width, int height, int channels) {
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return fail(VPT_E_IO, "deflate failed for %s", filename);
    if (zn > 0x7fffffffu) return fail(VPT_E_UNSUPPORTED, "%s: image too large for one IDAT chunk", filename);
    FILE* f = fopen(filename, "wb");
    if (!f) return fail(VPT_E_IO, "cannot write %s", filename);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    const uint32_t w = (uint32_t)width, h = (uint32_t)height;
    const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                              8, (uint8_t)(channels == 4 ? 6 : 2), 0, 0, 0};      // 8 bits per sample, colour type 2 (RGB) / 6 (RGBA), deflate, adaptive filtering, no interlace
    png_chunk(f, "IHDR", ihdr, sizeof(ihdr));
    png_chunk(f, "IDAT", z.data(), (size_t)zn);
    png_chunk(f, "IEND", nullptr, 0);
    const bool ok = ferror(f) == 0;
    fclose(f);
    return ok ? VPT_OK : fail(VPT_E_IO, "short write to %s", filename);
}
}  // namespace

int vpt_io_write_png(const char* filename, const unsigned int* display, int width, int height, int with_alpha) {
    if (!filename || !display || width <= 0 || height <= 0) return VPT_E_INVALID;
    const int ch = with_alpha ? 4 : 3;
    std::vector<uint8_t> raw(((size_t)width * ch + 1) * (size_t)height);
    uint8_t* o = raw.data();
    for (int y = 0; y < height; ++y) {
        *o++ = 0;                                                        // filter type 0 (None)
        for (int x = 0; x < width; ++x) {
            const unsigned int v = display[(size_t)y * width + x];       // 0xffRRGGBB (render_kernel.cu:2311)
            *o++ = (uint8_t)(v >> 16); *o++ = (uint8_t)(v >> 8); *o++ = (uint8_t)v;
            if (with_alpha) *o++ = (uint8_t)(v >> 24);
        }
    }
    return write_png8(filename, raw, width, height, ch);
}

// float3 / float4 image -> 8-bit PNG the way OpenImageIO converts FLOAT to UINT8 on write (save_texture_png(float3*), fileIO.cpp:110-123): clamp to [0, 1],
// scale by 255, round to nearest.  No tone curve: pass the raw buffer (display-referred) for a viewable image, the accumulation buffer for a clipped linear one.
int vpt_io_write_png_float(const char* filename, const float* pixels, int channels, int width, int height) {
    if (!filename || !pixels || (channels != 3 && channels != 4) || width <= 0 || height <= 0) return VPT_E_INVALID;
    std::vector<uint8_t> raw(((size_t)width * channels + 1) * (size_t)height);
    uint8_t* o = raw.data();
    for (int y = 0; y < height; ++y) {
        *o++ = 0;
        for (size_t i = 0; i < (size_t)width * channels; ++i) {
            const float v = pixels[(size_t)y * width * channels + i];
            const float c = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
            *o++ = (uint8_t)(c * 255.0f + 0.5f);
        }
    }
    return write_png8(filename, raw, width, height, channels);
}

}  // extern "C"
