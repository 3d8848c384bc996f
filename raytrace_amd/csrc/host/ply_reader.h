// A reader for PLY polygon files (ascii, binary_little_endian and binary_big_endian 1.0), for Shape "plymesh".
//
// The reference reads such files through the rply library (a dependency that is not part of its tree) with the
// callbacks of Enigma/TriangleMesh.cpp:98-174 and CreatePLYMesh (Enigma/TriangleMesh.cpp:791-931).  What reaches its
// TriangleMesh is restated here over a reader of the published format:
//  * element "vertex": x y z; nx ny nz if all three are there; the first complete pair of (u v), (s t),
//    (texture_u texture_v), (texture_s texture_t); every value goes through a double into a float;
//  * element "face": the list property "vertex_indices" (that name only); a face of three vertices is a triangle, one of
//    four the two triangles (0 1 2) and (3 0 2), any other length is skipped with a warning; an index outside
//    [0, vertices) makes the whole mesh an error; the scalar property "face_indices" is read along and not used;
//  * a file without vertices or faces is an error.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace RayTrace {

struct PlyMesh {
    std::vector<float> p, n, uv;      // 3, 3 (or empty), 2 (or empty) floats per vertex
    std::vector<int> indices;         // 3 per triangle
    std::vector<std::string> warnings;
};

class PlyReader {
public:
    static PlyMesh Read(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Couldn't open PLY file \"" + path + "\"");
        PlyReader r(f, path);
        r.Header();
        return r.Body();
    }

private:
    enum Type { I8, U8, I16, U16, I32, U32, F32, F64, BAD };
    struct Property { std::string name; bool list = false; Type count = BAD, value = BAD; };
    struct Element { std::string name; long instances = 0; std::vector<Property> props; };
    enum Format { ASCII, LITTLE, BIG };

    std::istream& m_in;
    std::string m_path;
    Format m_format = ASCII;
    std::vector<Element> m_elements;

    PlyReader(std::istream& in, const std::string& path) : m_in(in), m_path(path) {}

    [[noreturn]] void Fail(const std::string& what) const { throw std::runtime_error(m_path + ": " + what); }

    static Type TypeOf(const std::string& s) {
        if (s == "char" || s == "int8") return I8;
        if (s == "uchar" || s == "uint8") return U8;
        if (s == "short" || s == "int16") return I16;
        if (s == "ushort" || s == "uint16") return U16;
        if (s == "int" || s == "int32") return I32;
        if (s == "uint" || s == "uint32") return U32;
        if (s == "float" || s == "float32") return F32;
        if (s == "double" || s == "float64") return F64;
        return BAD;
    }
    static int SizeOf(Type t) { return t == I8 || t == U8 ? 1 : (t == I16 || t == U16 ? 2 : (t == F64 ? 8 : 4)); }

    void Header() {
        std::string line;
        if (!std::getline(m_in, line) || Trim(line) != "ply") Fail("Unable to read the header of PLY file (no \"ply\" magic)");
        bool haveFormat = false, ended = false;
        while (std::getline(m_in, line)) {
            std::istringstream ls(Trim(line));
            std::string word;
            if (!(ls >> word)) continue;
            if (word == "comment" || word == "obj_info") continue;
            if (word == "end_header") { ended = true; break; }
            if (word == "format") {
                std::string fmt, version;
                ls >> fmt >> version;
                if (fmt == "ascii") m_format = ASCII;
                else if (fmt == "binary_little_endian") m_format = LITTLE;
                else if (fmt == "binary_big_endian") m_format = BIG;
                else Fail("Unable to read the header of PLY file (format \"" + fmt + "\")");
                haveFormat = true;
            } else if (word == "element") {
                Element e;
                if (!(ls >> e.name >> e.instances) || e.instances < 0) Fail("Unable to read the header of PLY file (element line)");
                m_elements.push_back(e);
            } else if (word == "property") {
                if (m_elements.empty()) Fail("Unable to read the header of PLY file (property before any element)");
                Property pr;
                std::string t;
                ls >> t;
                if (t == "list") {
                    std::string ct, vt;
                    ls >> ct >> vt >> pr.name;
                    pr.list = true; pr.count = TypeOf(ct); pr.value = TypeOf(vt);
                    if (pr.count == BAD || pr.count == F32 || pr.count == F64) Fail("Unable to read the header of PLY file (list count type)");
                } else {
                    pr.value = TypeOf(t);
                    ls >> pr.name;
                }
                if (pr.value == BAD || pr.name.empty()) Fail("Unable to read the header of PLY file (property type \"" + t + "\")");
                m_elements.back().props.push_back(pr);
            } else Fail("Unable to read the header of PLY file (\"" + word + "\")");
        }
        if (!haveFormat || !ended) Fail("Unable to read the header of PLY file");
    }

    static std::string Trim(const std::string& s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }

    // one value of the body as the double rply hands to its callbacks
    double Value(Type t) {
        if (m_format == ASCII) {
            std::string tok;
            if (!(m_in >> tok)) Fail("unable to read the contents of PLY file");
            try {
                size_t used = 0;
                double v = std::stod(tok, &used);
                if (used != tok.size()) throw std::invalid_argument(tok);
                return (t == F32) ? (double)(float)v : (t == F64 ? v : (double)(long long)v);
            } catch (const std::exception&) { Fail("unable to read the contents of PLY file (\"" + tok + "\")"); }
        }
        unsigned char b[8];
        const int n = SizeOf(t);
        m_in.read(reinterpret_cast<char*>(b), n);
        if (!m_in) Fail("unable to read the contents of PLY file (truncated)");
        if (m_format == BIG) for (int i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
        case I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
        case U8: return b[0];
        case I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
        case U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
        case I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
        case U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
        case F32: { float v; std::memcpy(&v, b, 4); return v; }
        default: { double v; std::memcpy(&v, b, 8); return v; }
        }
    }

    PlyMesh Body() {
        PlyMesh mesh;
        long vertexCount = 0, faceCount = 0;
        for (const Element& e : m_elements) {
            if (e.name == "vertex") vertexCount = e.instances;
            else if (e.name == "face") faceCount = e.instances;
        }
        if (vertexCount == 0 || faceCount == 0) Fail("PLY file is invalid! No face/vertex elements found!");
        for (const Element& e : m_elements) {
            if (e.name == "vertex") {
                auto find = [&](const char* name) { for (size_t i = 0; i < e.props.size(); ++i) if (!e.props[i].list && e.props[i].name == name) return (int)i; return -1; };
                const int px = find("x"), py = find("y"), pz = find("z");
                if (px < 0 || py < 0 || pz < 0) Fail("Vertex coordinate property not found!");
                const int nx = find("nx"), ny = find("ny"), nz = find("nz");
                const bool haveN = nx >= 0 && ny >= 0 && nz >= 0;
                int tu = -1, tv = -1;
                for (const char* const* pair : {kUv[0], kUv[1], kUv[2], kUv[3]}) {
                    const int a = find(pair[0]), b = find(pair[1]);
                    if (a >= 0 && b >= 0) { tu = a; tv = b; break; }
                }
                mesh.p.resize((size_t)vertexCount * 3);
                if (haveN) mesh.n.resize((size_t)vertexCount * 3);
                if (tu >= 0) mesh.uv.resize((size_t)vertexCount * 2);
                for (long v = 0; v < vertexCount; ++v)
                    for (size_t k = 0; k < e.props.size(); ++k) {
                        const Property& pr = e.props[k];
                        if (pr.list) { const long len = (long)Value(pr.count); for (long j = 0; j < len; ++j) Value(pr.value); continue; }
                        const float val = (float)Value(pr.value);
                        const int i = (int)k;
                        if (i == px) mesh.p[3 * v] = val; else if (i == py) mesh.p[3 * v + 1] = val; else if (i == pz) mesh.p[3 * v + 2] = val;
                        if (haveN) { if (i == nx) mesh.n[3 * v] = val; else if (i == ny) mesh.n[3 * v + 1] = val; else if (i == nz) mesh.n[3 * v + 2] = val; }
                        if (tu >= 0) { if (i == tu) mesh.uv[2 * v] = val; else if (i == tv) mesh.uv[2 * v + 1] = val; }
                    }
            } else if (e.name == "face") {
                mesh.indices.reserve((size_t)faceCount * 3);
                long skipped = 0;
                for (long fc = 0; fc < e.instances; ++fc)
                    for (const Property& pr : e.props) {
                        if (!pr.list) { Value(pr.value); continue; }
                        const long len = (long)Value(pr.count);
                        if (len < 0) Fail("unable to read the contents of PLY file (negative list length)");
                        const bool wanted = pr.name == "vertex_indices";
                        int face[4] = {0, 0, 0, 0};
                        for (long j = 0; j < len; ++j) {
                            const double val = Value(pr.value);
                            if (!wanted || (len != 3 && len != 4)) continue;
                            const int idx = (int)val;
                            if (idx < 0 || idx >= vertexCount) {
                                std::ostringstream m;
                                m << "plymesh: Vertex reference " << idx << " is out of bounds! Valid range is [0.." << vertexCount << ")";
                                Fail(m.str());
                            }
                            face[j] = idx;
                        }
                        if (!wanted) continue;
                        if (len != 3 && len != 4) { ++skipped; continue; }
                        mesh.indices.insert(mesh.indices.end(), {face[0], face[1], face[2]});
                        if (len == 4) mesh.indices.insert(mesh.indices.end(), {face[3], face[0], face[2]});
                    }
                if (skipped) {
                    std::ostringstream m;
                    m << "plymesh: Ignoring " << skipped << " faces that are neither triangles nor quads";
                    mesh.warnings.push_back(m.str());
                }
            } else { // an element nobody asked for: read past it
                for (long k = 0; k < e.instances; ++k)
                    for (const Property& pr : e.props) {
                        if (pr.list) { const long len = (long)Value(pr.count); for (long j = 0; j < len; ++j) Value(pr.value); }
                        else Value(pr.value);
                    }
            }
        }
        return mesh;
    }

    static constexpr const char* kUv[4][2] = {{"u", "v"}, {"s", "t"}, {"texture_u", "texture_v"}, {"texture_s", "texture_t"}};
};

} // namespace RayTrace
