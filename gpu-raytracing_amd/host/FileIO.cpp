#include "FileIO.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace {

std::string BaseDirectory(const std::string& filename)
{
    const size_t found = filename.find_last_of("\\/");
    return found == std::string::npos ? std::string("") : filename.substr(0, found);
}

std::string JoinPath(const std::string& dir, const std::string& name) { return dir.empty() ? name : dir + "/" + name; }

bool FileExists(const std::string& filename)
{
    std::ifstream f(filename);
    return f.good();
}

std::vector<std::string> Tokens(const std::string& line)
{
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}

float ToFloat(const std::vector<std::string>& t, size_t i) { return i < t.size() ? (float)atof(t[i].c_str()) : 0.0f; }

struct Corner { int v = -1, t = -1, n = -1; };

// "v", "v/t", "v//n", "v/t/n"; 1-based, negative = relative to the end (FileIO.cpp:296-323)
Corner ParseCorner(const std::string& s, size_t nv, size_t nt, size_t nn)
{
    Corner c;
    int field = 0;
    size_t start = 0;
    for (size_t i = 0; i <= s.size(); i++) {
        if (i == s.size() || s[i] == '/') {
            if (i > start) {
                const int f = atoi(s.substr(start, i - start).c_str());
                const size_t n = field == 0 ? nv : (field == 1 ? nt : nn);
                const int idx = f < 0 ? f + (int)n : f - 1;
                if (field == 0) c.v = idx; else if (field == 1) c.t = idx; else if (field == 2) c.n = idx;
            }
            field++;
            start = i + 1;
        }
    }
    return c;
}

vec3 FlatNormal(const Triangle& tri) { return normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v1)); }  // FileIO.cpp:88-93

vec3 SetupLight(const std::string& obj_name, const AABB& aabb)  // FileIO.cpp:74-86
{
    vec3 result = aabb.Centre();
    const std::string name = JoinPath(BaseDirectory(obj_name), "light.txt");
    if (FILE* fp = fopen(name.c_str(), "r")) {
        float x, y, z;
        if (fscanf(fp, "%f %f %f", &x, &y, &z) == 3) result = make_vec3(x, y, z);
        fclose(fp);
    }
    return result;
}

// binary PPM (P6, maxval 255) -> RGBA8 with alpha 255
bool LoadPPM(const std::string& filename, Texture& tex)
{
    std::ifstream is(filename, std::ios::binary);
    if (!is.good()) return false;
    std::string magic;
    is >> magic;
    if (magic != "P6") return false;
    int vals[3], got = 0;
    while (got < 3 && is.good()) {
        is >> std::ws;
        if (is.peek() == '#') { std::string c; std::getline(is, c); continue; }
        is >> vals[got++];
    }
    if (got < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255) return false;
    is.get();   // the single whitespace after maxval
    std::vector<unsigned char> rgb((size_t)vals[0] * vals[1] * 3);
    is.read(reinterpret_cast<char*>(rgb.data()), (std::streamsize)rgb.size());
    if ((size_t)is.gcount() != rgb.size()) return false;
    tex.size_x[0] = vals[0];
    tex.size_y[0] = vals[1];
    tex.mips[0].resize((size_t)vals[0] * vals[1]);
    for (size_t i = 0; i < tex.mips[0].size(); i++)
        tex.mips[0][i] = rgb[3 * i] | (rgb[3 * i + 1] << 8) | (rgb[3 * i + 2] << 16) | 0xFF000000u;
    return true;
}

}  // namespace

uint32_t Texture::ReadTexel(int x, int y, int lod) const
{
    x = std::max(0, std::min(x, size_x[lod] - 1));
    y = std::max(0, std::min(y, size_y[lod] - 1));
    return mips[lod][(size_t)y * size_x[lod] + x];
}

void Texture::GenerateLODs()   // FileIO.cpp:121-150
{
    uint32_t lod = 0;
    while ((size_x[lod] > 1 || size_y[lod] > 1) && lod + 1 < (uint32_t)NUM_LODS) {
        size_x[lod + 1] = (size_x[lod] + 1) / 2;
        size_y[lod + 1] = (size_y[lod] + 1) / 2;
        mips[lod + 1].resize((size_t)size_x[lod + 1] * size_y[lod + 1]);
        for (int j = 0; j < size_y[lod + 1]; j++)
            for (int i = 0; i < size_x[lod + 1]; i++) {
                const uint32_t t[4] = {ReadTexel(i * 2, j * 2, lod), ReadTexel(i * 2 + 1, j * 2, lod),
                                       ReadTexel(i * 2, j * 2 + 1, lod), ReadTexel(i * 2 + 1, j * 2 + 1, lod)};
                uint32_t out = 0;
                for (int c = 0; c < 4; c++) {
                    const float sum = (((float)((t[0] >> (8 * c)) & 255u) + (float)((t[1] >> (8 * c)) & 255u)) +
                                       (float)((t[2] >> (8 * c)) & 255u)) + (float)((t[3] >> (8 * c)) & 255u);
                    out |= ((uint32_t)(sum * 0.25f) & 255u) << (8 * c);
                }
                mips[lod + 1][(size_t)j * size_x[lod + 1] + i] = out;
            }
        lod++;
    }
    max_lod = lod;
}

int32_t Library::AddTexture(const std::string& filename)
{
    auto it = name_to_tex.find(filename);
    if (it != name_to_tex.end()) return (int32_t)it->second;
    printf("Loading %s\n", filename.c_str());
    Texture tex;
    tex.name = filename;
    if (!LoadPPM(filename, tex)) {
        fprintf(stderr, "warning: %s is not a binary PPM (the only texture format this build decodes); material keeps no texture\n", filename.c_str());
        return -1;
    }
    tex.GenerateLODs();
    name_to_tex[filename] = (uint32_t)textures.size();
    textures.push_back(std::move(tex));
    return (int32_t)textures.size() - 1;
}

Library LoadMTLFromFile(const std::string& filename)  // FileIO.cpp:222-287
{
    Library library;
    printf("Loading MTL file: %s\n", filename.c_str());
    std::ifstream fs(filename);
    std::string line;
    while (std::getline(fs, line)) {
        const auto t = Tokens(line);
        if (t.empty()) continue;
        if (t[0] == "newmtl" && t.size() > 1) {
            library.AddMaterial(t[1]);
        } else if (library.materials.empty()) {
            continue;
        } else if ((t[0] == "Ka" || t[0] == "Kd" || t[0] == "Ks") && t.size() > 1) {
            const vec3 v = t.size() >= 4 ? make_vec3(ToFloat(t, 1), ToFloat(t, 2), ToFloat(t, 3)) : make_vec3(ToFloat(t, 1));
            Material& m = library.materials.back();
            if (t[0] == "Ka") m.ambient = v; else if (t[0] == "Kd") m.diffuse = v; else m.specular = v;
        } else if (t[0] == "Ns" && t.size() > 1) {
            library.materials.back().specular_exp = ToFloat(t, 1);
        } else if (t[0] == "map_Kd" && t.size() > 1) {
            library.materials.back().texture_file = JoinPath(BaseDirectory(filename), t[1]);
            library.materials.back().texture = library.AddTexture(library.materials.back().texture_file);
        } else if (t[0] == "bump" && t.size() > 1) {
            library.materials.back().bump_file = JoinPath(BaseDirectory(filename), t[1]);
            library.materials.back().bump = library.AddTexture(library.materials.back().bump_file);
        } else if (t[0] == "map_Disp" && t.size() > 1) {
            library.materials.back().disp_file = JoinPath(BaseDirectory(filename), t[1]);
            library.materials.back().disp = library.AddTexture(library.materials.back().disp_file);
        }
    }
    return library;
}

Scene LoadOBJFromFile(const std::string& filename)  // FileIO.cpp:327-457
{
    std::ifstream fs(filename);
    if (!fs.good()) throw std::runtime_error("Can't open OBJ file " + filename);
    Scene scene;
    std::vector<vec3> verts, normals;
    std::vector<std::pair<float, float>> uvs;
    int32_t current_material = -1;
    std::string line;
    while (std::getline(fs, line)) {
        const auto t = Tokens(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "mtllib" && t.size() > 1) {
            std::string mtl = t[1];
            if (!FileExists(mtl)) mtl = JoinPath(BaseDirectory(filename), mtl);
            scene.library = LoadMTLFromFile(mtl);
        } else if (t[0] == "usemtl" && t.size() > 1) {
            current_material = scene.library.GetMaterialId(t[1]);
        } else if (t[0] == "v") {
            verts.push_back(make_vec3(ToFloat(t, 1), ToFloat(t, 2), ToFloat(t, 3)));
        } else if (t[0] == "vt") {
            uvs.emplace_back(ToFloat(t, 1), ToFloat(t, 2));
        } else if (t[0] == "vn") {
            normals.push_back(make_vec3(ToFloat(t, 1), ToFloat(t, 2), ToFloat(t, 3)));
        } else if (t[0] == "f") {
            std::vector<Corner> c;
            for (size_t i = 1; i < t.size(); i++) c.push_back(ParseCorner(t[i], verts.size(), uvs.size(), normals.size()));
            for (size_t i = 2; i < c.size(); i++) {
                const Corner k[3] = {c[0], c[i - 1], c[i]};
                bool ok = true;
                for (const Corner& q : k) ok = ok && q.v >= 0 && (size_t)q.v < verts.size();
                if (!ok) continue;  // the reference indexes out of bounds here
                const Triangle tri{verts[k[0].v], verts[k[1].v], verts[k[2].v]};
                scene.triangles.push_back(tri);
                Attributes a{};
                a.material_id = current_material;
                const vec3 flat = FlatNormal(tri);
                for (int j = 0; j < 3; j++) {
                    const bool has_t = k[j].t >= 0 && (size_t)k[j].t < uvs.size();
                    a.uv[j][0] = has_t ? uvs[k[j].t].first : 0.0f;
                    a.uv[j][1] = has_t ? uvs[k[j].t].second : 0.0f;
                    a.normal[j] = (k[j].n >= 0 && (size_t)k[j].n < normals.size()) ? normals[k[j].n] : flat;
                }
                scene.attributes.push_back(a);
            }
        }
    }
    printf("Geometry\n  faces:        %u\n  verts:        %u\n", (unsigned)scene.triangles.size(), (unsigned)verts.size());
    scene.aabb = {make_vec3(FLT_MAX), make_vec3(-FLT_MAX)};
    for (const Triangle& t : scene.triangles) {
        scene.aabb = Combine(scene.aabb, t.v0);
        scene.aabb = Combine(scene.aabb, t.v1);
        scene.aabb = Combine(scene.aabb, t.v2);
    }
    printf("  aabb: (%f %f %f %f %f %f)\n", scene.aabb.min.x, scene.aabb.min.y, scene.aabb.min.z, scene.aabb.max.x,
           scene.aabb.max.y, scene.aabb.max.z);
    scene.light = SetupLight(filename, scene.aabb);
    printf("  light: %f %f %f\n", scene.light.x, scene.light.y, scene.light.z);
    return scene;
}
