/*
 * kkd_mesh_io.cpp -- on-disk mesh formats on the fast path: the step right before the hot path.
 *
 * Restates, host-only and without Mitsuba's Stream classes:
 *   TriMesh::loadCompressed / readHeader / readOffset / readOffsetDictionary   src/librender/trimesh.cpp:174-337
 *   TriMesh::serialize(Stream*)  (the .serialized writer, version 4)             src/librender/trimesh.cpp:1060-1108
 *   SerializedMesh (toWorld, winding flip for mirroring transforms)              src/shapes/serialized.cpp:150-206
 *   PLYLoader (vertex / face callbacks, quad split, sRGB vertex colours)        src/shapes/ply.cpp:96-330, 326-470
 *   WavefrontOBJ (statement loop, corner forms, createMesh)                     src/shapes/obj.cpp:165-349, 373-392, 586-717
 *
 * The file is memory-mapped and the deflate stream is inflated straight into the destination arrays (no intermediate
 * copy of a >= 10 M-triangle scan); the arrays are what kkd_scene_desc points at.
 */
#include "kkd_host.h"
#include <zlib.h>
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace kkd { std::string g_globalError; }

namespace {

using kkd::g_globalError;

constexpr uint16_t kHeader = 0x041C, kV3 = 0x0003, kV4 = 0x0004;          /* trimesh.cpp:33-35 */
enum : uint32_t { EHasNormals = 0x0001, EHasTexcoords = 0x0002, EHasColors = 0x0008, EFaceNormals = 0x0010,
                  ESinglePrecision = 0x1000, EDoublePrecision = 0x2000 };   /* trimesh.cpp:88-96 */

int ioFail(const std::string &msg) { g_globalError = msg; return KKD_ERR_INVALID; }

struct MappedFile {
    const uint8_t *p = nullptr; size_t size = 0; int fd = -1;
    bool open(const char *path, std::string &err) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open \"") + path + "\": " + std::strerror(errno); return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { err = std::string("\"") + path + "\" is empty or unreadable"; return false; }
        size = (size_t)st.st_size;
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { err = std::string("mmap of \"") + path + "\" failed"; return false; }
        madvise(m, size, MADV_SEQUENTIAL);
        p = (const uint8_t *)m;
        return true;
    }
    ~MappedFile() { if (p) munmap((void *)p, size); if (fd >= 0) ::close(fd); }
};

template <typename T> T loadLE(const uint8_t *p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }   /* host is little endian (gfx950 boxes are x86-64) */

/* ZStream read side (libcore/zstream.cpp:29-46: zlib wrapper, windowBits 15), pulling from the mapped file */
struct Inflater {
    z_stream zs; bool ok = false;
    Inflater(const uint8_t *src, size_t n) {
        std::memset(&zs, 0, sizeof(zs));
        ok = inflateInit2(&zs, 15) == Z_OK;
        zs.next_in = const_cast<Bytef *>(src);
        remaining = n; feed();
    }
    ~Inflater() { if (ok) inflateEnd(&zs); }
    size_t remaining = 0;
    void feed() { const size_t c = std::min<size_t>(remaining, 1u << 30); zs.avail_in = (uInt)c; remaining -= c; }
    bool read(void *dst, size_t n) {
        uint8_t *out = (uint8_t *)dst;
        while (n > 0) {
            const size_t c = std::min<size_t>(n, 1u << 30);
            zs.next_out = out; zs.avail_out = (uInt)c;
            while (zs.avail_out > 0) {
                if (zs.avail_in == 0) { if (remaining == 0) return false; feed(); }
                const int rc = inflate(&zs, Z_NO_FLUSH);
                if (rc == Z_STREAM_END) { if (zs.avail_out > 0) return false; break; }
                if (rc != Z_OK) return false;
            }
            out += c; n -= c;
        }
        return true;
    }
};

/* readHelper, trimesh.cpp:146-172: file precision -> float */
bool readFloats(Inflater &in, bool fileDouble, float *dst, size_t count) {
    if (!fileDouble) return in.read(dst, count * sizeof(float));
    std::vector<double> tmp(std::min<size_t>(count, 1u << 20));
    while (count > 0) {
        const size_t c = std::min(count, tmp.size());
        if (!in.read(tmp.data(), c * sizeof(double))) return false;
        for (size_t i = 0; i < c; ++i) dst[i] = (float)tmp[i];
        dst += c; count -= c;
    }
    return true;
}

struct Xform {
    bool identity = true; float m[16]; float invT[9]; float det = 1.f;
    void set(const float *toWorld) {
        identity = true;
        if (!toWorld) return;
        static const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        if (std::memcmp(toWorld, I, sizeof(I)) == 0) return;
        identity = false;
        std::memcpy(m, toWorld, sizeof(m));
        const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
        const double dt = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        det = (float)dt;
        const double id = dt != 0.0 ? 1.0 / dt : 0.0;
        /* inverse of the upper 3x3; a normal is multiplied by its transpose (Transform::operator()(Normal)) */
        const double inv[9] = { (e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                                (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                                (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id };
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) invT[r * 3 + k] = (float)inv[k * 3 + r];
    }
    void point(float *p) const {      /* Transform::operator()(Point): affine part + homogeneous divide when w != 1 */
        const float x = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
        const float y = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
        const float z = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
        const float w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
        if (w == 1.0f) { p[0] = x; p[1] = y; p[2] = z; } else { p[0] = x / w; p[1] = y / w; p[2] = z / w; }
    }
    void normal(float *n) const {     /* normalize(objectToWorld(normal)), serialized.cpp:191-193 / ply.cpp:229 */
        const float x = invT[0] * n[0] + invT[1] * n[1] + invT[2] * n[2];
        const float y = invT[3] * n[0] + invT[4] * n[1] + invT[5] * n[2];
        const float z = invT[6] * n[0] + invT[7] * n[1] + invT[8] * n[2];
        const float l = std::sqrt(x * x + y * y + z * z);
        if (l > 0.f) { n[0] = x / l; n[1] = y / l; n[2] = z / l; } else { n[0] = x; n[1] = y; n[2] = z; }
    }
};

void resetMesh(kkd_mesh *m) { std::memset(m, 0, sizeof(*m)); }

void finishAabb(kkd_mesh *m) {
    for (int a = 0; a < 3; ++a) { m->aabb_min[a] = INFINITY; m->aabb_max[a] = -INFINITY; }
    for (uint64_t v = 0; v < m->n_verts; ++v)
        for (int a = 0; a < 3; ++a) {
            m->aabb_min[a] = std::min(m->aabb_min[a], m->positions[3 * v + a]);
            m->aabb_max[a] = std::max(m->aabb_max[a], m->positions[3 * v + a]);
        }
}

template <typename T> T *allocArray(size_t n) { return (T *)std::malloc(std::max<size_t>(n, 1) * sizeof(T)); }

/* ---- .serialized dictionary, trimesh.cpp:269-337 ---- */
bool readDictionary(const MappedFile &f, uint16_t version, std::vector<uint64_t> &offsets) {
    if (f.size < 8) return false;
    const uint32_t count = loadLE<uint32_t>(f.p + f.size - 4);
    const size_t minSize = 4 + (size_t)count * (2 * 2 + 4 + 1 + 2 * 8 + 3 * 4 + 3 * 4);   /* trimesh.cpp:303-309 */
    if (count == 0 || f.size < minSize) return false;
    offsets.resize(count);
    if (version == kV4) {
        const size_t base = f.size - 8 * (size_t)count - 4;
        for (uint32_t i = 0; i < count; ++i) offsets[i] = loadLE<uint64_t>(f.p + base + 8 * (size_t)i);
    } else {
        const size_t base = f.size - 4 * ((size_t)count + 1);
        for (uint32_t i = 0; i < count; ++i) offsets[i] = loadLE<uint32_t>(f.p + base + 4 * (size_t)i);
    }
    return true;
}

int readHeader(const uint8_t *p, size_t avail, uint16_t &version) {     /* trimesh.cpp:251-267 */
    if (avail < 4) return ioFail("serialized mesh: truncated header");
    const uint16_t format = loadLE<uint16_t>(p);
    if (format == 0x1C04) return ioFail("serialized mesh: geometry file generated by an old version of Mitsuba");
    if (format != kHeader) return ioFail("serialized mesh: invalid file format");
    version = loadLE<uint16_t>(p + 2);
    if (version != kV3 && version != kV4) return ioFail("serialized mesh: incompatible file version");
    return KKD_OK;
}

/* ---- PLY ---- */
enum PlyType { P_I8, P_U8, P_I16, P_U16, P_I32, P_U32, P_F32, P_F64, P_BAD };
const int kPlySize[] = { 1, 1, 2, 2, 4, 4, 4, 8, 0 };
PlyType plyType(const std::string &s) {
    if (s == "char" || s == "int8") return P_I8;
    if (s == "uchar" || s == "uint8") return P_U8;
    if (s == "short" || s == "int16") return P_I16;
    if (s == "ushort" || s == "uint16") return P_U16;
    if (s == "int" || s == "int32") return P_I32;
    if (s == "uint" || s == "uint32") return P_U32;
    if (s == "float" || s == "float32") return P_F32;
    if (s == "double" || s == "float64") return P_F64;
    return P_BAD;
}
struct PlyProp { std::string name; bool list = false; PlyType type = P_BAD, sizeType = P_BAD; int slot = -1; };
struct PlyElem { std::string name; uint64_t count = 0; std::vector<PlyProp> props; };

/* slots a vertex property can feed (ply.cpp:326-392): only float32 properties, colours also as uint8 */
enum { S_X, S_Y, S_Z, S_NX, S_NY, S_NZ, S_U, S_V, S_R, S_G, S_B, S_COUNT };
int vertexSlot(const PlyProp &p) {
    const std::string &n = p.name;
    int s = -1;
    if (n == "x") s = S_X; else if (n == "y") s = S_Y; else if (n == "z") s = S_Z;
    else if (n == "nx") s = S_NX; else if (n == "ny") s = S_NY; else if (n == "nz") s = S_NZ;
    else if (n == "u" || n == "texture_u" || n == "s") s = S_U;
    else if (n == "v" || n == "texture_v" || n == "t") s = S_V;
    else if (n == "diffuse_red" || n == "red") s = S_R;
    else if (n == "diffuse_green" || n == "green") s = S_G;
    else if (n == "diffuse_blue" || n == "blue") s = S_B;
    if (s < 0) return -1;
    if (p.type == P_F32) return s;
    if (p.type == P_U8 && s >= S_R) return s;
    return -1;      /* libply finds no callback for other types: the property is parsed and dropped */
}

struct PlyReader {
    const uint8_t *p, *end; int mode;    /* 0 ascii, 1 little, 2 big */
    bool fail = false;
    bool token(const char *&b, const char *&e) {
        while (p < end && std::isspace(*p)) ++p;
        if (p >= end) { fail = true; return false; }
        b = (const char *)p;
        while (p < end && !std::isspace(*p)) ++p;
        e = (const char *)p;
        return true;
    }
    double scalar(PlyType t) {
        if (mode == 0) {
            const char *b, *e;
            if (!token(b, e)) return 0;
            char buf[64]; const size_t n = std::min<size_t>(e - b, 63); std::memcpy(buf, b, n); buf[n] = 0;
            return t >= P_F32 ? std::strtod(buf, nullptr) : (double)std::strtoll(buf, nullptr, 10);
        }
        const int sz = kPlySize[t];
        if (end - p < sz) { fail = true; return 0; }
        uint8_t tmp[8];
        if (mode == 1) std::memcpy(tmp, p, sz); else for (int i = 0; i < sz; ++i) tmp[i] = p[sz - 1 - i];
        p += sz;
        switch (t) {
            case P_I8: return (double)(int8_t)tmp[0];
            case P_U8: return (double)tmp[0];
            case P_I16: return (double)loadLE<int16_t>(tmp);
            case P_U16: return (double)loadLE<uint16_t>(tmp);
            case P_I32: return (double)loadLE<int32_t>(tmp);
            case P_U32: return (double)loadLE<uint32_t>(tmp);
            case P_F32: return (double)loadLE<float>(tmp);
            default: return loadLE<double>(tmp);
        }
    }
};

float fromSRGB(float v) {      /* ply.cpp:214-219 */
    if (v <= 0.04045f) return v / 12.92f;
    return std::pow((v + 0.055f) / (float)(1.0 + 0.055), 2.4f);
}

} /* namespace */

extern "C" {

void kkd_mesh_free(kkd_mesh *m) {
    if (!m) return;
    std::free(m->positions); std::free(m->normals); std::free(m->texcoords); std::free(m->colors); std::free(m->indices);
    resetMesh(m);
}

int kkd_mesh_count_serialized(const char *path, uint32_t *count) {
    if (!path || !count) return ioFail("kkd_mesh_count_serialized: null argument");
    MappedFile f; std::string err;
    if (!f.open(path, err)) return ioFail(err);
    uint16_t version; int rc = readHeader(f.p, f.size, version);
    if (rc != KKD_OK) return rc;
    std::vector<uint64_t> offsets;
    *count = readDictionary(f, version, offsets) ? (uint32_t)offsets.size() : 1u;   /* serialized.cpp:232-236: no dictionary = one mesh at 0 */
    return KKD_OK;
}

int kkd_mesh_load_serialized(const char *path, int32_t shape_index, const float *to_world, kkd_mesh *out) {
    if (!path || !out) return ioFail("kkd_mesh_load_serialized: null argument");
    resetMesh(out);
    if (shape_index < 0) return ioFail("serialized mesh: shape index must be nonnegative");
    MappedFile f; std::string err;
    if (!f.open(path, err)) return ioFail(err);
    uint16_t version; int rc = readHeader(f.p, f.size, version);
    if (rc != KKD_OK) return rc;
    size_t offset = 0;
    std::vector<uint64_t> offsets;
    if (readDictionary(f, version, offsets)) {
        if ((size_t)shape_index >= offsets.size()) return ioFail("serialized mesh: shape index is out of range");
        offset = (size_t)offsets[shape_index];
    } else if (shape_index != 0) return ioFail("serialized mesh: shape index is out of range (file has no dictionary)");
    if (offset + 4 > f.size) return ioFail("serialized mesh: offset beyond the end of the file");
    rc = readHeader(f.p + offset, f.size - offset, version);      /* every sub-mesh repeats the header, trimesh.cpp:183-187 */
    if (rc != KKD_OK) return rc;
    Inflater in(f.p + offset + 4, f.size - offset - 4);
    if (!in.ok) return ioFail("serialized mesh: could not initialise zlib");
    uint32_t flags;
    if (!in.read(&flags, 4)) return ioFail("serialized mesh: truncated stream");
    if (version == kV4) {                                           /* Stream::readString: bytes up to a NUL, stream.cpp:416-426 */
        size_t n = 0; char c;
        do { if (!in.read(&c, 1)) return ioFail("serialized mesh: truncated name"); if (c && n + 1 < sizeof(out->name)) out->name[n++] = c; } while (c);
    }
    uint64_t nv, nt;
    if (!in.read(&nv, 8) || !in.read(&nt, 8)) return ioFail("serialized mesh: truncated stream");
    if (nv == 0 || nt == 0 || nv > (1ull << 32) || nt > (1ull << 32)) return ioFail("serialized mesh: implausible vertex / triangle count");
    const bool fileDouble = (flags & EDoublePrecision) != 0;
    out->n_verts = nv; out->n_tris = nt;
    out->flags = (flags & EFaceNormals) ? KKD_MESH_FACE_NORMALS : 0u;
    out->positions = allocArray<float>(3 * nv);
    out->indices = allocArray<uint32_t>(3 * nt);
    if (flags & EHasNormals) out->normals = allocArray<float>(3 * nv);
    if (flags & EHasTexcoords) out->texcoords = allocArray<float>(2 * nv);
    if (flags & EHasColors) out->colors = allocArray<float>(3 * nv);
    bool ok = out->positions && out->indices && (!(flags & EHasNormals) || out->normals) && (!(flags & EHasTexcoords) || out->texcoords) &&
              (!(flags & EHasColors) || out->colors);
    if (!ok) { kkd_mesh_free(out); g_globalError = "serialized mesh: out of memory"; return KKD_ERR_NOMEM; }
    ok = readFloats(in, fileDouble, out->positions, 3 * nv);
    if (ok && out->normals) ok = readFloats(in, fileDouble, out->normals, 3 * nv);
    if (ok && out->texcoords) ok = readFloats(in, fileDouble, out->texcoords, 2 * nv);
    if (ok && out->colors) ok = readFloats(in, fileDouble, out->colors, 3 * nv);
    if (ok) ok = in.read(out->indices, 3 * nt * sizeof(uint32_t));
    if (!ok) { kkd_mesh_free(out); return ioFail("serialized mesh: truncated or corrupt deflate stream"); }
    for (uint64_t i = 0; i < 3 * nt; ++i) if (out->indices[i] >= nv) { kkd_mesh_free(out); return ioFail("serialized mesh: vertex index out of range"); }
    Xform x; x.set(to_world);
    if (!x.identity) {                                               /* serialized.cpp:183-201 */
        for (uint64_t v = 0; v < nv; ++v) x.point(out->positions + 3 * v);
        if (out->normals) for (uint64_t v = 0; v < nv; ++v) x.normal(out->normals + 3 * v);
        if (x.det < 0.f) for (uint64_t t = 0; t < nt; ++t) std::swap(out->indices[3 * t], out->indices[3 * t + 1]);
    }
    finishAabb(out);
    return KKD_OK;
}

/* TriMesh::serialize(Stream*), trimesh.cpp:1060-1108, for n meshes + the version-4 dictionary a multi-mesh file ends
 * with (64-bit offsets, then the mesh count; layout read back by readOffsetDictionary, trimesh.cpp:296-337). */
int kkd_mesh_save_serialized(const char *path, const kkd_mesh *meshes, uint32_t n_meshes) {
    if (!path || !meshes || n_meshes == 0) return ioFail("kkd_mesh_save_serialized: null argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return ioFail(std::string("cannot create \"") + path + "\"");
    std::vector<uint64_t> offsets;
    bool ok = true;
    for (uint32_t k = 0; k < n_meshes && ok; ++k) {
        const kkd_mesh &m = meshes[k];
        offsets.push_back((uint64_t)std::ftell(f));
        const uint16_t hdr[2] = { kHeader, kV4 };
        ok = std::fwrite(hdr, 2, 2, f) == 2;
        uint32_t flags = ESinglePrecision;
        if (m.normals) flags |= EHasNormals;
        if (m.texcoords) flags |= EHasTexcoords;
        if (m.colors) flags |= EHasColors;
        if (m.flags & KKD_MESH_FACE_NORMALS) flags |= EFaceNormals;
        z_stream zs; std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; break; }
        std::vector<uint8_t> buf(1 << 20);
        auto put = [&](const void *src, size_t n, int flush) {
            const uint8_t *s = (const uint8_t *)src;
            do {
                const size_t c = std::min<size_t>(n, 1u << 30);
                zs.next_in = const_cast<Bytef *>(s); zs.avail_in = (uInt)c; s += c; n -= c;
                const int fl = n == 0 ? flush : Z_NO_FLUSH;
                do {
                    zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
                    const int rc = deflate(&zs, fl);
                    if (rc == Z_STREAM_ERROR) { ok = false; return; }
                    const size_t have = buf.size() - zs.avail_out;
                    if (have && std::fwrite(buf.data(), 1, have, f) != have) { ok = false; return; }
                } while (zs.avail_out == 0);
            } while (n > 0);
        };
        const size_t nameLen = strnlen(m.name, sizeof(m.name));
        std::string name(m.name, nameLen);
        put(&flags, 4, Z_NO_FLUSH);
        put(name.c_str(), nameLen + 1, Z_NO_FLUSH);                  /* writeString: characters + NUL, stream.cpp:222-224 */
        put(&m.n_verts, 8, Z_NO_FLUSH); put(&m.n_tris, 8, Z_NO_FLUSH);
        put(m.positions, 3 * m.n_verts * 4, Z_NO_FLUSH);
        if (m.normals) put(m.normals, 3 * m.n_verts * 4, Z_NO_FLUSH);
        if (m.texcoords) put(m.texcoords, 2 * m.n_verts * 4, Z_NO_FLUSH);
        if (m.colors) put(m.colors, 3 * m.n_verts * 4, Z_NO_FLUSH);
        put(m.indices, 3 * m.n_tris * 4, Z_FINISH);
        deflateEnd(&zs);
    }
    if (ok) {
        ok = std::fwrite(offsets.data(), 8, offsets.size(), f) == offsets.size();
        ok = ok && std::fwrite(&n_meshes, 4, 1, f) == 1;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KKD_OK : ioFail("kkd_mesh_save_serialized: write failed");
}

int kkd_mesh_load_ply(const char *path, const float *to_world, int32_t srgb, kkd_mesh *out) {
    if (!path || !out) return ioFail("kkd_mesh_load_ply: null argument");
    resetMesh(out);
    MappedFile f; std::string err;
    if (!f.open(path, err)) return ioFail(err);
    /* ---- header ---- */
    const uint8_t *p = f.p, *end = f.p + f.size;
    auto line = [&](std::string &s) { s.clear(); while (p < end && *p != '\n') s.push_back((char)*p++); if (p < end) ++p; if (!s.empty() && s.back() == '\r') s.pop_back(); return !s.empty() || p < end; };
    std::string ln;
    if (!line(ln) || ln != "ply") return ioFail("PLY: missing magic");
    int mode = -1; std::vector<PlyElem> elems; bool headerDone = false;
    while (line(ln)) {
        char a[64] = "", b[64] = "", c[64] = "", d[64] = "", e[64] = "";
        const int n = std::sscanf(ln.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, e);
        if (n <= 0) continue;
        const std::string key(a);
        if (key == "format") { const std::string fm(b); mode = fm == "ascii" ? 0 : fm == "binary_little_endian" ? 1 : fm == "binary_big_endian" ? 2 : -1; }
        else if (key == "comment" || key == "obj_info") continue;
        else if (key == "element") { PlyElem el; el.name = b; el.count = std::strtoull(c, nullptr, 10); elems.push_back(el); }
        else if (key == "property") {
            if (elems.empty()) return ioFail("PLY: property before element");
            PlyProp pr;
            if (std::string(b) == "list") { pr.list = true; pr.sizeType = plyType(c); pr.type = plyType(d); pr.name = e; }
            else { pr.type = plyType(b); pr.name = c; }
            if (pr.type == P_BAD || (pr.list && pr.sizeType == P_BAD)) return ioFail("PLY: unknown property type");
            elems.back().props.push_back(pr);
        } else if (key == "end_header") { headerDone = true; break; }
    }
    if (!headerDone || mode < 0) return ioFail("PLY: malformed header");
    /* ---- body ---- */
    Xform x; x.set(to_world);
    PlyReader rd{ p, end, mode };
    uint64_t faceCount = 0, triCount = 0;
    bool hasN = false, hasUV = false, hasC = false;
    for (PlyElem &el : elems) {
        if (el.name == "vertex") {
            if (el.count == 0 || el.count > (1ull << 32)) return ioFail("PLY: implausible vertex count");
            out->n_verts = el.count;
            for (PlyProp &pr : el.props) {
                pr.slot = pr.list ? -1 : vertexSlot(pr);
                if (pr.slot >= S_NX && pr.slot <= S_NZ) hasN = true;
                if (pr.slot == S_U || pr.slot == S_V) hasUV = true;
                if (pr.slot >= S_R) hasC = true;
            }
            out->positions = allocArray<float>(3 * el.count);
            if (hasN) out->normals = allocArray<float>(3 * el.count);
            if (hasUV) out->texcoords = allocArray<float>(2 * el.count);
            if (hasC) out->colors = allocArray<float>(3 * el.count);
            if (!out->positions || (hasN && !out->normals) || (hasUV && !out->texcoords) || (hasC && !out->colors)) { kkd_mesh_free(out); g_globalError = "PLY: out of memory"; return KKD_ERR_NOMEM; }
            float cur[S_COUNT] = { 0 };                               /* the loader's m_position / m_normal / m_uv / m_red.. members persist */
            for (uint64_t v = 0; v < el.count; ++v) {
                for (const PlyProp &pr : el.props) {
                    if (pr.list) { const uint64_t c = (uint64_t)rd.scalar(pr.sizeType); for (uint64_t i = 0; i < c && !rd.fail; ++i) rd.scalar(pr.type); continue; }
                    const double val = rd.scalar(pr.type);
                    if (pr.slot >= 0) cur[pr.slot] = (pr.type == P_U8) ? (float)val / 255.0f : (float)val;
                }
                if (rd.fail) { kkd_mesh_free(out); return ioFail("PLY: truncated vertex data"); }
                float *P = out->positions + 3 * v; P[0] = cur[S_X]; P[1] = cur[S_Y]; P[2] = cur[S_Z];
                if (!x.identity) x.point(P);
                if (hasN) { float *N = out->normals + 3 * v; N[0] = cur[S_NX]; N[1] = cur[S_NY]; N[2] = cur[S_NZ];
                            if (!x.identity) x.normal(N); else { const float l = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]); if (l > 0.f) { N[0] /= l; N[1] /= l; N[2] /= l; } } }
                if (hasUV) { out->texcoords[2 * v] = cur[S_U]; out->texcoords[2 * v + 1] = cur[S_V]; }
                if (hasC) for (int k = 0; k < 3; ++k) out->colors[3 * v + k] = srgb ? fromSRGB(cur[S_R + k]) : cur[S_R + k];
            }
        } else if (el.name == "face") {
            if (el.count > (1ull << 31)) return ioFail("PLY: implausible face count");
            faceCount = el.count;
            out->indices = allocArray<uint32_t>(6 * el.count);       /* m_triangles = new Triangle[m_faceCount*2], ply.cpp:181 */
            if (!out->indices) { kkd_mesh_free(out); g_globalError = "PLY: out of memory"; return KKD_ERR_NOMEM; }
            for (uint64_t fc = 0; fc < el.count; ++fc) {
                for (const PlyProp &pr : el.props) {
                    if (!pr.list) { rd.scalar(pr.type); continue; }
                    const uint64_t c = (uint64_t)rd.scalar(pr.sizeType);
                    /* callbacks exist for size uint8 / uint32 and index int32 / uint32 only (ply.cpp:394-470) */
                    const bool wanted = (pr.name == "vertex_indices" || pr.name == "vertex_index") && (pr.sizeType == P_U8 || pr.sizeType == P_U32) &&
                                        (pr.type == P_I32 || pr.type == P_U32);
                    if (!wanted) { for (uint64_t i = 0; i < c && !rd.fail; ++i) rd.scalar(pr.type); continue; }
                    if (c != 3 && c != 4) { kkd_mesh_free(out); return ioFail("PLY: only triangle and quad-based PLY meshes are supported"); }
                    uint32_t face[4] = { 0, 0, 0, 0 };
                    for (uint64_t i = 0; i < c; ++i) {
                        const double v = rd.scalar(pr.type);
                        if (v < 0 || v >= (double)out->n_verts) { kkd_mesh_free(out); return ioFail("PLY: vertex index out of range"); }
                        face[i] = (uint32_t)v;
                    }
                    uint32_t *t = out->indices + 3 * triCount;
                    t[0] = face[0]; t[1] = face[1]; t[2] = face[2]; ++triCount;
                    if (c == 4) { t[3] = face[3]; t[4] = face[0]; t[5] = face[2]; ++triCount; }   /* ply.cpp:303-306 */
                }
                if (rd.fail) { kkd_mesh_free(out); return ioFail("PLY: truncated face data"); }
            }
        } else {
            for (uint64_t i = 0; i < el.count && !rd.fail; ++i)
                for (const PlyProp &pr : el.props) {
                    if (!pr.list) { rd.scalar(pr.type); continue; }
                    const uint64_t c = (uint64_t)rd.scalar(pr.sizeType);
                    for (uint64_t k = 0; k < c && !rd.fail; ++k) rd.scalar(pr.type);
                }
        }
    }
    (void)faceCount;
    out->n_tris = triCount;
    if (out->n_tris == 0 || out->n_verts == 0) { kkd_mesh_free(out); return ioFail("PLY: no triangles or vertices found"); }   /* ply.cpp:116-117 */
    const size_t stem0 = std::string(path).find_last_of('/') == std::string::npos ? 0 : std::string(path).find_last_of('/') + 1;
    std::string stem = std::string(path).substr(stem0);
    if (stem.find_last_of('.') != std::string::npos) stem = stem.substr(0, stem.find_last_of('.'));
    std::snprintf(out->name, sizeof(out->name), "%s", stem.c_str());  /* m_name = filePath.stem(), ply.cpp:99 */
    finishAabb(out);
    return KKD_OK;
}

} /* extern "C" */

namespace {

/* ---- Wavefront OBJ: the obj plugin's reader (src/shapes/obj.cpp:165-349, 373-392, 586-717) -------------------------------
 * One pass over the file as the plugin's constructor makes it: "v" / "vn" / "vt" grow three global lists, "f" appends
 * triangles (n-gons as fans), "g" and "usemtl" close the current geometry when it holds triangles, and every geometry that
 * is closed becomes one TriMesh through createMesh (corner = (position, normal, uv) AFTER toWorld, equal corners merged
 * through an ordered map).  Quirks kept because a drop-in must name and split meshes the same way: both naming conventions
 * of "g" (name before or after the geometry), "_<index>" suffixes for repeated names, substr(1) / substr(6) of the raw line
 * for the names, a face with fewer than three corners repeating its last token, backslash line continuation. */
struct ObjTri { int p[3], n[3], uv[3]; };
struct ObjGeom { std::string name, material; std::vector<ObjTri> tris; size_t nV = 0, nN = 0, nT = 0; /* list sizes when it was closed */ };
struct ObjFile { std::vector<float> v, vn, vt; std::vector<ObjGeom> geoms; bool eofMesh = false; /* the last geometry was closed by the end of the file */ };

std::string objTrim(const std::string &s) {            /* mitsuba::trim, src/libcore/util.cpp: blanks, tabs, CR, LF */
    const char *ws = " \t\r\n";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}

struct ObjLines {                                      /* fetch_line, obj.cpp:165-187 */
    const char *p, *end;
    bool raw(std::string &line) { if (p >= end) return false; const char *q = (const char *)std::memchr(p, '\n', (size_t)(end - p)); if (!q) q = end; line.assign(p, q); p = q < end ? q + 1 : end; return true; }
    bool fetch(std::string &line) {
        if (!raw(line)) return false;
        if (line.empty()) return true;
        int last = (int)line.size() - 1;
        while (last >= 0 && (line[last] == '\r' || line[last] == '\n' || line[last] == '\t' || line[last] == ' ')) --last;
        if (last >= 0 && line[last] == '\\') { std::string next; fetch(next); line = line.substr(0, (size_t)last) + next; }
        else line.resize((size_t)(last + 1));
        return true;
    }
};

struct ObjTokens {                                     /* std::istringstream >> std::string */
    const std::string &s; size_t pos = 0;
    explicit ObjTokens(const std::string &str) : s(str) {}
    bool next(std::string &tok) {
        while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos;
        if (pos >= s.size()) return false;
        const size_t a = pos;
        while (pos < s.size() && !std::isspace((unsigned char)s[pos])) ++pos;
        tok.assign(s, a, pos - a);
        return true;
    }
    float number() { std::string t; if (!next(t)) return 0.f; return std::strtof(t.c_str(), nullptr); }
};

bool objCorner(ObjTri &t, int i, const std::string &str) {      /* parse(), obj.cpp:373-392; tokenize() drops empty tokens */
    std::vector<std::string> tok; size_t a = 0;
    while (a <= str.size()) {
        const size_t b = str.find('/', a);
        const std::string part = str.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (!part.empty()) tok.push_back(part);
        if (b == std::string::npos) break;
        a = b + 1;
    }
    if (tok.size() == 1) t.p[i] = std::atoi(tok[0].c_str());
    else if (tok.size() == 2) {
        t.p[i] = std::atoi(tok[0].c_str());
        if (str.find("//") == std::string::npos) t.uv[i] = std::atoi(tok[1].c_str()); else t.n[i] = std::atoi(tok[1].c_str());
    } else if (tok.size() == 3) { t.p[i] = std::atoi(tok[0].c_str()); t.uv[i] = std::atoi(tok[1].c_str()); t.n[i] = std::atoi(tok[2].c_str()); }
    else return false;
    return true;
}

int parseObj(const char *path, uint32_t flags, ObjFile &of) {
    MappedFile f; std::string err;
    if (!f.open(path, err)) return ioFail(err);
    const bool collapse = (flags & KKD_OBJ_COLLAPSE) != 0, flipV = (flags & KKD_OBJ_KEEP_TEXCOORD_V) == 0;
    std::string stem = path;
    if (stem.find_last_of('/') != std::string::npos) stem = stem.substr(stem.find_last_of('/') + 1);
    if (stem.find_last_of('.') != std::string::npos && stem.find_last_of('.') > 0) stem = stem.substr(0, stem.find_last_of('.'));   /* m_name = path.stem() */
    ObjLines lines{ (const char *)f.p, (const char *)f.p + f.size };
    std::string line, buf, name = stem, material;
    std::vector<std::string> seen;                    /* geomNames */
    auto taken = [&](const std::string &n) { return std::find(seen.begin(), seen.end(), n) != seen.end(); };
    std::vector<ObjTri> tris;
    int geomIndex = 0; bool nameBeforeGeometry = false;
    auto close = [&](const std::string &as) {          /* createMesh's arguments at this point of the file */
        ObjGeom g; g.name = as; g.material = material; g.tris.swap(tris); g.nV = of.v.size() / 3; g.nN = of.vn.size() / 3; g.nT = of.vt.size() / 2;
        of.geoms.push_back(std::move(g)); tris.clear();
    };
    while (lines.fetch(line)) {
        ObjTokens tk(line);
        if (!tk.next(buf)) continue;
        if (buf == "v") { for (int k = 0; k < 3; ++k) of.v.push_back(tk.number()); }
        else if (buf == "vn") { for (int k = 0; k < 3; ++k) of.vn.push_back(tk.number()); }
        else if (buf == "g" && !collapse) {
            std::string target; const std::string newName = objTrim(line.substr(1));
            target = nameBeforeGeometry ? name : newName;            /* obj.cpp:262-269: two conventions for where the name goes */
            if (!tris.empty()) {
                if (taken(target)) target += "_" + std::to_string(geomIndex);
                geomIndex += 1; seen.push_back(target);
                close(target);
            } else nameBeforeGeometry = true;
            name = newName;
        } else if (buf == "usemtl") {
            if (!tris.empty() && !collapse) {
                if (taken(name)) name += "_" + std::to_string(geomIndex);
                geomIndex += 1; seen.push_back(name);
                close(name);
                name = stem;
            }
            material = objTrim(line.substr(6));
        } else if (buf == "vt") { const float u = tk.number(); float v = tk.number(); if (flipV) v = 1 - v; of.vt.push_back(u); of.vt.push_back(v); }
        else if (buf == "f") {
            std::string tmp; ObjTri t; std::memset(&t, 0, sizeof(t));
            for (int k = 0; k < 3; ++k) { tk.next(tmp); if (!objCorner(t, k, tmp)) return ioFail("Invalid OBJ face format!"); }   /* a missing token leaves tmp as it was */
            tris.push_back(t);
            while (tk.next(tmp)) {                                   /* n-gons as fans, assuming a convex shape */
                t.p[1] = t.p[2]; t.uv[1] = t.uv[2]; t.n[1] = t.n[2];
                if (!objCorner(t, 2, tmp)) return ioFail("Invalid OBJ face format!");
                tris.push_back(t);
            }
        }
    }
    if (taken(name)) name = stem + "_" + std::to_string(geomIndex);
    if (!tris.empty()) { close(name); of.eofMesh = true; }
    return KKD_OK;
}

int buildObjMesh(const ObjFile &of, const ObjGeom &g, const float *to_world, kkd_mesh *out) {     /* createMesh, obj.cpp:610-717 */
    Xform x; x.set(to_world);
    struct Key { float k[8]; bool operator<(const Key &o) const { for (int i = 0; i < 8; ++i) { if (k[i] < o.k[i]) return true; if (k[i] > o.k[i]) return false; } return false; } };
    std::map<Key, uint32_t> merged; std::vector<Key> verts; std::vector<uint32_t> idx; idx.reserve(3 * g.tris.size());
    bool hasN = false, hasUV = false;
    char msg[128];
    for (const ObjTri &t : g.tris)
        for (int j = 0; j < 3; ++j) {
            int vi = t.p[j], ni = t.n[j], ti = t.uv[j];
            if (vi < 0) vi += (int)g.nV + 1;
            if (ni < 0) ni += (int)g.nN + 1;
            if (ti < 0) ti += (int)g.nT + 1;
            if (vi > (int)g.nV || vi <= 0) { std::snprintf(msg, sizeof(msg), "Out of bounds: tried to access vertex %i (max: %i)", vi, (int)g.nV); return ioFail(msg); }
            Key key; std::memset(&key, 0, sizeof(key));
            key.k[0] = of.v[3 * (size_t)(vi - 1)]; key.k[1] = of.v[3 * (size_t)(vi - 1) + 1]; key.k[2] = of.v[3 * (size_t)(vi - 1) + 2];
            if (!x.identity) x.point(key.k);
            if (ni != 0) {
                if (ni > (int)g.nN || ni < 0) { std::snprintf(msg, sizeof(msg), "Out of bounds: tried to access normal %i (max: %i)", ni, (int)g.nN); return ioFail(msg); }
                float *n = key.k + 3; n[0] = of.vn[3 * (size_t)(ni - 1)]; n[1] = of.vn[3 * (size_t)(ni - 1) + 1]; n[2] = of.vn[3 * (size_t)(ni - 1) + 2];
                if (!x.identity) x.normal(n);                        /* objectToWorld(Normal), then normalize unless zero */
                else { const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); if (l > 0.f) { n[0] /= l; n[1] /= l; n[2] /= l; } }
                hasN = true;
            }
            if (ti != 0) {
                if (ti > (int)g.nT || ti < 0) { std::snprintf(msg, sizeof(msg), "Out of bounds: tried to access uv %i (max: %i)", ti, (int)g.nT); return ioFail(msg); }
                key.k[6] = of.vt[2 * (size_t)(ti - 1)]; key.k[7] = of.vt[2 * (size_t)(ti - 1) + 1];
                hasUV = true;
            }
            auto it = merged.find(key);
            uint32_t id;
            if (it != merged.end()) id = it->second; else { id = (uint32_t)verts.size(); merged.emplace(key, id); verts.push_back(key); }
            idx.push_back(id);
        }
    out->n_verts = verts.size(); out->n_tris = g.tris.size();
    out->positions = allocArray<float>(3 * verts.size()); out->indices = allocArray<uint32_t>(idx.size());
    if (hasN) out->normals = allocArray<float>(3 * verts.size());
    if (hasUV) out->texcoords = allocArray<float>(2 * verts.size());
    if (!out->positions || !out->indices || (hasN && !out->normals) || (hasUV && !out->texcoords)) { kkd_mesh_free(out); g_globalError = "OBJ: out of memory"; return KKD_ERR_NOMEM; }
    for (size_t v = 0; v < verts.size(); ++v) {
        std::memcpy(out->positions + 3 * v, verts[v].k, 12);
        if (hasN) std::memcpy(out->normals + 3 * v, verts[v].k + 3, 12);
        if (hasUV) std::memcpy(out->texcoords + 2 * v, verts[v].k + 6, 8);
    }
    std::memcpy(out->indices, idx.data(), idx.size() * sizeof(uint32_t));
    std::snprintf(out->name, sizeof(out->name), "%s", g.name.c_str());
    finishAabb(out);
    return KKD_OK;
}

} /* namespace */

extern "C" {

int kkd_mesh_count_obj(const char *path, uint32_t flags, uint32_t *count, uint32_t *eof_mesh) {
    if (!path || !count) return ioFail("kkd_mesh_count_obj: null argument");
    ObjFile of;
    const int rc = parseObj(path, flags, of);
    if (rc != KKD_OK) return rc;
    *count = (uint32_t)of.geoms.size();
    if (eof_mesh) *eof_mesh = of.eofMesh ? 1u : 0u;
    return KKD_OK;
}

int kkd_mesh_load_obj(const char *path, uint32_t mesh_index, const float *to_world, uint32_t flags, kkd_mesh *out, char *material, uint32_t material_capacity) {
    if (!path || !out) return ioFail("kkd_mesh_load_obj: null argument");
    resetMesh(out);
    ObjFile of;
    int rc = parseObj(path, flags, of);
    if (rc != KKD_OK) return rc;
    if (mesh_index >= of.geoms.size()) return ioFail("kkd_mesh_load_obj: mesh index " + std::to_string(mesh_index) + " out of range (the file holds " + std::to_string(of.geoms.size()) + ")");
    rc = buildObjMesh(of, of.geoms[mesh_index], to_world, out);
    if (rc != KKD_OK) return rc;
    if (material && material_capacity) std::snprintf(material, material_capacity, "%s", of.geoms[mesh_index].material.c_str());
    return KKD_OK;
}

/* TriMesh::rebuildTopology (src/librender/trimesh.cpp:472-610), what the 'maxSmoothAngle' parameter of the ply / serialized /
 * obj plugins triggers (src/shapes/ply.cpp:124-128, serialized.cpp:204-208, obj.cpp:336-342): vertices are keyed by (position,
 * uv, colour); the triangle corners that share a key are clustered greedily by their face normals -- a corner joins the
 * cluster of the first unclustered corner n1 of its key when n1 == n2 or dot(n1, n2) > cos(maxAngle) -- and every cluster
 * becomes one vertex of the new mesh; stored normals are dropped (the caller regenerates them as TriMesh::configure does).
 * The reference walks a std::multimap<Vertex, TopoData>; equal keys keep insertion order there (triangle by triangle,
 * corner by corner), which a stable sort of the corner list reproduces. */
int kkd_mesh_rebuild_topology(kkd_mesh *m, float max_angle_deg) {
    if (!m || !m->positions || !m->indices || m->n_tris == 0) return ioFail("kkd_mesh_rebuild_topology: empty mesh");
    for (uint64_t i = 0; i < 3 * m->n_tris; ++i) if (m->indices[i] >= m->n_verts) return ioFail("kkd_mesh_rebuild_topology: vertex index out of range");
    const float dpThresh = std::cos(max_angle_deg * (float)(M_PI / 180.0));            /* degToRad, trimesh.cpp:475 */
    const bool hasUV = m->texcoords != nullptr, hasC = m->colors != nullptr;
    struct Corner { float key[8]; uint32_t tri; bool clustered; };
    const size_t nt = (size_t)m->n_tris;
    std::vector<Corner> corners(3 * nt);
    std::vector<float> faceN(3 * nt);
    for (size_t i = 0; i < nt; ++i) {
        const uint32_t *t = m->indices + 3 * i;
        for (int j = 0; j < 3; ++j) {
            Corner &c = corners[3 * i + j];
            const float *p = m->positions + 3 * (size_t)t[j];
            c.key[0] = p[0]; c.key[1] = p[1]; c.key[2] = p[2];
            c.key[3] = hasUV ? m->texcoords[2 * (size_t)t[j]] : 0.f; c.key[4] = hasUV ? m->texcoords[2 * (size_t)t[j] + 1] : 0.f;
            for (int k = 0; k < 3; ++k) c.key[5 + k] = hasC ? m->colors[3 * (size_t)t[j] + k] : 0.f;
            c.tri = (uint32_t)i; c.clustered = false;
        }
        const float *v0 = m->positions + 3 * (size_t)t[0], *v1 = m->positions + 3 * (size_t)t[1], *v2 = m->positions + 3 * (size_t)t[2];
        const float a[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, b[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
        float n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
        const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (l > 0x1p-128f) { n[0] /= l; n[1] /= l; n[2] /= l; } else n[0] = n[1] = n[2] = 0.f;   /* RCPOVERFLOW_FLT: degenerate triangle */
        faceN[3 * i] = n[0]; faceN[3 * i + 1] = n[1]; faceN[3 * i + 2] = n[2];
    }
    auto less = [](const Corner &x, const Corner &y) {                                   /* vertex_key_order, trimesh.cpp:437-460 */
        for (int k = 0; k < 8; ++k) { if (x.key[k] < y.key[k]) return true; if (x.key[k] > y.key[k]) return false; }
        return false;
    };
    std::stable_sort(corners.begin(), corners.end(), less);
    std::vector<float> newPos, newUV, newCol;
    newPos.reserve(3 * (size_t)m->n_verts);
    std::vector<uint32_t> newIdx(3 * nt, 0xFFFFFFFFu);
    for (size_t start = 0; start < corners.size();) {
        size_t end = start + 1;
        while (end < corners.size() && !less(corners[start], corners[end])) ++end;        /* equal range of the key */
        for (size_t i2 = start; i2 < end; ++i2) {                                         /* greedy clustering of normals, :543-573 */
            if (corners[i2].clustered) continue;
            const float *n1 = &faceN[3 * (size_t)corners[i2].tri];
            const uint32_t vertexIdx = (uint32_t)(newPos.size() / 3);
            const float *key = corners[i2].key;
            newPos.insert(newPos.end(), key, key + 3);
            if (hasUV) newUV.insert(newUV.end(), key + 3, key + 5);
            if (hasC) newCol.insert(newCol.end(), key + 5, key + 8);
            for (size_t i3 = i2; i3 < end; ++i3) {
                Corner &c2 = corners[i3];
                if (c2.clustered) continue;
                const float *n2 = &faceN[3 * (size_t)c2.tri];
                const bool same = n1[0] == n2[0] && n1[1] == n2[1] && n1[2] == n2[2];
                if (same || n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2] > dpThresh) {
                    const uint32_t *t = m->indices + 3 * (size_t)c2.tri;
                    for (int k = 0; k < 3; ++k) {                                         /* every corner of that triangle AT this position, :566-569 */
                        const float *p = m->positions + 3 * (size_t)t[k];
                        if (p[0] == key[0] && p[1] == key[1] && p[2] == key[2]) newIdx[3 * (size_t)c2.tri + k] = vertexIdx;
                    }
                    c2.clustered = true;
                }
            }
        }
        start = end;
    }
    for (uint32_t v : newIdx) if (v == 0xFFFFFFFFu) return ioFail("kkd_mesh_rebuild_topology: a corner was left without a vertex");   /* Assert, :578-580 */
    const size_t nv = newPos.size() / 3;
    float *pos = allocArray<float>(3 * nv), *uv = hasUV ? allocArray<float>(2 * nv) : nullptr, *col = hasC ? allocArray<float>(3 * nv) : nullptr;
    if (!pos || (hasUV && !uv) || (hasC && !col)) { std::free(pos); std::free(uv); std::free(col); g_globalError = "kkd_mesh_rebuild_topology: out of memory"; return KKD_ERR_NOMEM; }
    std::memcpy(pos, newPos.data(), newPos.size() * sizeof(float));
    if (hasUV) std::memcpy(uv, newUV.data(), newUV.size() * sizeof(float));
    if (hasC) std::memcpy(col, newCol.data(), newCol.size() * sizeof(float));
    std::memcpy(m->indices, newIdx.data(), newIdx.size() * sizeof(uint32_t));
    std::free(m->positions); std::free(m->normals); std::free(m->texcoords); std::free(m->colors);
    m->positions = pos; m->normals = nullptr; m->texcoords = uv; m->colors = col; m->n_verts = nv;
    return KKD_OK;
}

} /* extern "C" */
