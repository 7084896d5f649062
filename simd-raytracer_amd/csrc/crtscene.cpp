// .crtscene reader: a small JSON DOM + the mapping of io/json/loader.hpp:46-265 onto rtk_scene.
// The reference parses with simdjson (absent offline); numbers take the same double -> float path
// (loader.hpp:9-17) via strtod, which is correctly rounded like simdjson's parser.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "rtk_internal.hpp"

namespace rtk {

int finish_mesh(HostMesh &m, std::string &err);   // scene.cpp

namespace {

struct JValue;
using JPtr = std::unique_ptr<JValue>;

struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;

    const JValue *get(const char *key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj) if (kv.first == key) return kv.second.get();
        return nullptr;
    }
};

struct Parser {
    const char *p, *end;
    std::string err;

    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }

    bool string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {  // keep BMP code points as UTF-8; scene files only use ASCII
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            const char c = p[i];
                            cp <<= 4;
                            if (c >= '0' && c <= '9') cp |= unsigned(c - '0');
                            else if (c >= 'a' && c <= 'f') cp |= unsigned(c - 'a' + 10);
                            else if (c >= 'A' && c <= 'F') cp |= unsigned(c - 'A' + 10);
                            else return fail("bad \\u escape");
                        }
                        p += 4;
                        if (cp < 0x80) out += char(cp);
                        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
                        else { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p; break;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }

    bool value(JValue &v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of input");
        const char c = *p;
        if (c == '{') {
            v.kind = JValue::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JPtr child(new JValue);
                if (!value(*child, depth + 1)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JValue::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JPtr child(new JValue);
                if (!value(*child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JValue::String; return string(v.str); }
        if (end - p >= 4 && !std::strncmp(p, "true", 4)) { v.kind = JValue::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && !std::strncmp(p, "false", 5)) { v.kind = JValue::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && !std::strncmp(p, "null", 4)) { v.kind = JValue::Null; p += 4; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            char *stop = nullptr;
            errno = 0;
            v.num = std::strtod(p, &stop);
            if (stop == p) return fail("bad number");
            v.kind = JValue::Number; p = stop;
            return true;
        }
        return fail("unexpected character");
    }
};

struct Fail { int code; std::string msg; };

const JValue &need(const JValue *parent, const char *key, JValue::Kind kind, const char *what) {
    const JValue *v = parent ? parent->get(key) : nullptr;
    if (!v) throw Fail{RTK_ERR_PARSE, std::string("missing key '") + key + "' in " + what};
    if (v->kind != kind) throw Fail{RTK_ERR_PARSE, std::string("key '") + key + "' in " + what + " has the wrong type"};
    return *v;
}

float f32(const JValue &v, const char *what) {                         // loader.hpp:9-17
    if (v.kind != JValue::Number) throw Fail{RTK_ERR_PARSE, std::string(what) + ": expected a number"};
    return static_cast<float>(v.num);
}

void floats(const JValue &arr, size_t n, float *out, const char *what) { // load_vec3 / load_mat3 / load_color
    if (arr.arr.size() < n) throw Fail{RTK_ERR_PARSE, std::string(what) + ": array too short"};
    for (size_t i = 0; i < n; ++i) out[i] = f32(*arr.arr[i], what);
}

bool file_exists(const std::string &p) {
    if (std::FILE *f = std::fopen(p.c_str(), "rb")) { std::fclose(f); return true; }
    return false;
}

// bitmap.hpp:15 hands `file_path` to stbi_load as it stands, i.e. relative to the working directory, which the reference's README
// asks to be its project root (README.md:32-35).  Here: as given if that exists; else below the nearest ancestor directory of
// the scene file that holds it, also with leading components dropped (scenes/hw12/textures/x.jpg next to a copy of scenes/).
std::string resolve_texture_path(const std::string &scene_path, const std::string &file_path) {
    if (file_exists(file_path) || file_path.empty() || file_path[0] == '/') return file_path;
    std::string dir = scene_path;
    for (;;) {
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash);
        const std::string base = dir.empty() ? (slash == std::string::npos ? std::string(".") : std::string("/")) : dir;
        size_t from = 0;
        for (;;) {
            const std::string cand = base + "/" + file_path.substr(from);
            if (file_exists(cand)) return cand;
            const size_t next = file_path.find('/', from);
            if (next == std::string::npos) break;
            from = next + 1;
        }
        if (slash == std::string::npos || dir.empty()) return file_path;
    }
}

int64_t uint_of(const JValue &v, const char *what) {
    if (v.kind != JValue::Number || v.num < 0 || v.num != static_cast<double>(static_cast<int64_t>(v.num)))
        throw Fail{RTK_ERR_PARSE, std::string(what) + ": expected an unsigned integer"};
    return static_cast<int64_t>(v.num);
}

}  // namespace

int scene_from_crtscene(const char *path, rtk_scene &out, std::string &err) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return RTK_ERR_IO; }
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    std::fclose(f);

    Parser ps{text.data(), text.data() + text.size(), {}};
    JValue root;
    if (!ps.value(root, 0)) { err = "JSON: " + ps.err; return RTK_ERR_PARSE; }
    ps.ws();
    if (ps.p != ps.end) { err = "JSON: trailing characters"; return RTK_ERR_PARSE; }

    try {
        // load_settings, loader.hpp:46-60
        const JValue &settings = need(&root, "settings", JValue::Object, "scene");
        floats(need(&settings, "background_color", JValue::Array, "settings"), 3, out.background, "background_color");
        const JValue &img = need(&settings, "image_settings", JValue::Object, "settings");
        out.width = static_cast<int32_t>(uint_of(need(&img, "width", JValue::Number, "image_settings"), "width"));
        out.height = static_cast<int32_t>(uint_of(need(&img, "height", JValue::Number, "image_settings"), "height"));
        out.bucket_size = 64;
        if (const JValue *bs = img.get("bucket_size")) {
            if (bs->kind == JValue::Number && bs->num >= 0 && bs->num == static_cast<double>(static_cast<int64_t>(bs->num)))
                out.bucket_size = static_cast<int32_t>(bs->num);
        }
        // load_camera, loader.hpp:62-68
        const JValue &cam = need(&root, "camera", JValue::Object, "scene");
        floats(need(&cam, "position", JValue::Array, "camera"), 3, out.cam_pos, "camera.position");
        floats(need(&cam, "matrix", JValue::Array, "camera"), 9, out.cam_mat, "camera.matrix");
        // lights, loader.hpp:245-247 (a missing key is an error there too)
        out.lights.clear();
        for (const JPtr &l : need(&root, "lights", JValue::Array, "scene").arr) {
            DevLight dl;
            floats(need(l.get(), "position", JValue::Array, "light"), 3, dl.pos, "light.position");
            dl.intensity = f32(need(l.get(), "intensity", JValue::Number, "light"), "light.intensity");
            out.lights.push_back(dl);
        }
        // load_texture, loader.hpp:78-106; scene.textures is keyed by name there (loader.hpp:249-253), by index here
        out.textures.clear();
        std::vector<std::string> texture_names;
        std::vector<Fail> texture_fail;          // a bitmap file that could not be decoded: an error once a material uses it
        out.tex_pixels.clear();
        if (const JValue *texs = root.get("textures")) {
            if (texs->kind == JValue::Array) for (const JPtr &t : texs->arr) {
                DevTexture dt;
                std::memset(&dt, 0, sizeof(dt));
                const std::string &type = need(t.get(), "type", JValue::String, "texture").str;
                Fail load_fail{RTK_OK, {}};
                if (type == "albedo") {
                    dt.kind = RTK_TEX_ALBEDO;
                    floats(need(t.get(), "albedo", JValue::Array, "texture"), 3, dt.a, "texture.albedo");
                } else if (type == "edges") {
                    dt.kind = RTK_TEX_EDGES;
                    floats(need(t.get(), "edge_color", JValue::Array, "texture"), 3, dt.a, "texture.edge_color");
                    floats(need(t.get(), "inner_color", JValue::Array, "texture"), 3, dt.b, "texture.inner_color");
                    dt.param = f32(need(t.get(), "edge_width", JValue::Number, "texture"), "texture.edge_width");
                } else if (type == "checker") {
                    dt.kind = RTK_TEX_CHECKER;
                    floats(need(t.get(), "color_A", JValue::Array, "texture"), 3, dt.a, "texture.color_A");
                    floats(need(t.get(), "color_B", JValue::Array, "texture"), 3, dt.b, "texture.color_B");
                    dt.param = f32(need(t.get(), "square_size", JValue::Number, "texture"), "texture.square_size");
                } else if (type == "bitmap") {
                    dt.kind = RTK_TEX_BITMAP;                                 // loader.hpp:97-101, texture/bitmap.hpp:11-44
                    const std::string &file_path = need(t.get(), "file_path", JValue::String, "texture").str;
                    int w = 0, h = 0, ch = 0;
                    std::vector<uint8_t> px;
                    std::string jerr;
                    const int jrc = load_bitmap_file(resolve_texture_path(path, file_path), w, h, ch, px, jerr);
                    if (jrc != RTK_OK || ch != 3) dt.kind = RTK_TEX_ALBEDO;     // placeholder; using it is the error (below)
                    if (jrc != RTK_OK) load_fail = Fail{jrc, jerr};
                    else if (ch != 3) load_fail = Fail{RTK_ERR_UNSUPPORTED, "bitmap texture '" + file_path + "' is not a 3-channel image (bitmap.hpp:26-28 reads three)"};
                    else {
                        dt.bmp[0] = w; dt.bmp[1] = h; dt.bmp[2] = static_cast<int32_t>(out.tex_pixels.size());
                        out.tex_pixels.insert(out.tex_pixels.end(), px.begin(), px.end());
                    }
                } else {
                    throw Fail{RTK_ERR_INVALID, "texture type unknown"};
                }
                texture_names.push_back(need(t.get(), "name", JValue::String, "texture").str);
                texture_fail.push_back(load_fail);
                out.textures.push_back(dt);
            }
        }
        // load_material, loader.hpp:108-147
        out.materials.clear();
        for (const JPtr &m : need(&root, "materials", JValue::Array, "scene").arr) {
            DevMaterial dm;
            std::memset(&dm, 0, sizeof(dm));
            dm.ior = 1.0f;
            dm.texture = -1;
            const std::string &type = need(m.get(), "type", JValue::String, "material").str;
            if (type == "diffuse") {
                const JValue *alb = m->get("albedo");
                if (!alb) throw Fail{RTK_ERR_PARSE, "missing key 'albedo' in material"};
                if (alb->kind == JValue::String) {                    // texture_material, loader.hpp:120-125
                    dm.kind = RTK_MAT_TEXTURE;
                    dm.texture = -1;
                    // scene.textures is a map filled with emplace (loader.hpp:249-253): of two textures with one name the FIRST stays
                    for (size_t ti = 0; ti < texture_names.size(); ++ti)
                        if (texture_names[ti] == alb->str) { dm.texture = static_cast<int32_t>(ti); break; }
                    if (dm.texture < 0) throw Fail{RTK_ERR_INVALID, "material refers to unknown texture '" + alb->str + "'"};
                    if (texture_fail[static_cast<size_t>(dm.texture)].code != RTK_OK) throw texture_fail[static_cast<size_t>(dm.texture)];
                } else if (alb->kind == JValue::Array) {
                    dm.kind = RTK_MAT_DIFFUSE;
                    floats(*alb, 3, dm.albedo, "material.albedo");
                } else {
                    throw Fail{RTK_ERR_INVALID, "albedo neither array nor string"};
                }
            } else if (type == "reflective") {
                dm.kind = RTK_MAT_REFLECTIVE;
                floats(need(m.get(), "albedo", JValue::Array, "material"), 3, dm.albedo, "material.albedo");
            } else if (type == "refractive") {
                dm.kind = RTK_MAT_REFRACTIVE;
                dm.ior = f32(need(m.get(), "ior", JValue::Number, "material"), "material.ior");
            } else if (type == "constant") {
                dm.kind = RTK_MAT_CONSTANT;
                floats(need(m.get(), "albedo", JValue::Array, "material"), 3, dm.albedo, "material.albedo");
            } else {
                throw Fail{RTK_ERR_INVALID, "material type unknown"};
            }
            dm.smooth = need(m.get(), "smooth_shading", JValue::Bool, "material").b ? 1 : 0;
            out.materials.push_back(dm);
        }
        // load_mesh, loader.hpp:149-233
        out.meshes.clear();
        out.n_vertices = out.n_triangles = 0;
        for (const JPtr &o : need(&root, "objects", JValue::Array, "scene").arr) {
            HostMesh mesh;
            const int64_t mi = uint_of(need(o.get(), "material_index", JValue::Number, "object"), "material_index");
            if (mi >= static_cast<int64_t>(out.materials.size())) throw Fail{RTK_ERR_INVALID, "material_index out of range"};
            mesh.material = static_cast<int32_t>(mi);
            const JValue &vs = need(o.get(), "vertices", JValue::Array, "object");
            if (vs.arr.size() % 3) throw Fail{RTK_ERR_INVALID, "vertex coordinates not multiple of 3"};
            mesh.vertices.resize(vs.arr.size() / 3);
            for (size_t i = 0; i < mesh.vertices.size(); ++i)
                mesh.vertices[i] = {f32(*vs.arr[i * 3], "vertices"), f32(*vs.arr[i * 3 + 1], "vertices"), f32(*vs.arr[i * 3 + 2], "vertices")};
            if (const JValue *uv = o->get("uvs")) {                      // loader.hpp:173-192: (u, v, ignored) triples
                if (uv->kind == JValue::Array) {
                    if (uv->arr.size() % 3) throw Fail{RTK_ERR_INVALID, "uv coordinates not multiple of 3"};
                    mesh.uvs.resize(uv->arr.size() / 3 * 2);
                    for (size_t i = 0; i < uv->arr.size() / 3; ++i) {
                        mesh.uvs[i * 2] = f32(*uv->arr[i * 3], "uvs");
                        mesh.uvs[i * 2 + 1] = f32(*uv->arr[i * 3 + 1], "uvs");
                    }
                    if (mesh.uvs.size() != mesh.vertices.size() * 2) {
                        if (mesh.uvs.empty()) mesh.uvs.clear();
                        else if (mesh.uvs.size() < mesh.vertices.size() * 2) throw Fail{RTK_ERR_INVALID, "fewer uvs than vertices"};
                        else mesh.uvs.resize(mesh.vertices.size() * 2);
                    }
                }
            }
            const JValue &ts = need(o.get(), "triangles", JValue::Array, "object");
            if (ts.arr.size() % 3) throw Fail{RTK_ERR_INVALID, "triangle indices not multiple of 3"};
            mesh.indices.resize(ts.arr.size());
            for (size_t i = 0; i < ts.arr.size(); ++i) mesh.indices[i] = static_cast<uint32_t>(uint_of(*ts.arr[i], "triangles"));
            std::string merr;
            const int rc = finish_mesh(mesh, merr);
            if (rc != RTK_OK) throw Fail{rc, merr};
            out.n_vertices += static_cast<int32_t>(mesh.vertices.size());
            out.n_triangles += static_cast<int32_t>(mesh.indices.size() / 3);
            out.meshes.push_back(std::move(mesh));
        }
    } catch (const Fail &fl) {
        err = std::string(path) + ": " + fl.msg;
        return fl.code;
    }
    return RTK_OK;
}

}  // namespace rtk
