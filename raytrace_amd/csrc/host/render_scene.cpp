// Host driver written against the mirrored reference interface (enigma_host.h): loads a flat scene
// file, builds Film / PerspectiveCamera / SobolSampler / Scene / GpuPathIntegrator (or GpuDirectLightingIntegrator) the way
// RenderOptions::MakeIntegrator + MakeScene do (Enigma/Api.cpp:1170-1254), calls
// integrator->Render(scene) like Enigma/Main.h:245, and writes the film's RGB as raw floats.
//
//   enigma_render <scene.bin> <out.f32>
// scene.bin (little endian): "ESC1", then int32 {xres, yres, spp, maxDepth, integrator (PathIntegrator: 0 uniform / 1 power
//   light distribution; DirectLightingIntegrator: 2 strategy all / 3 strategy one),
//   nVerts, nTris, nMaterials}, float {rrThreshold, lensRadius, focalDistance}, float[16] rasterToCamera,
//   float[16] cameraToWorld, float[nVerts*3], int32[nTris*3], int32[nTris] material,
//   int32[nTris] emissive flag, float[nTris*3] Le, EnigmaMaterial[nMaterials].
#include "enigma_host.h"

#include <cstring>
#include <fstream>
#include <iostream>

using namespace RayTrace;

template <class T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: enigma_render scene.bin out.f32\n"; return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    char magic[4];
    rd(f, magic, 4);
    if (!f || memcmp(magic, "ESC1", 4) != 0) { std::cerr << "bad scene file\n"; return 2; }
    int32_t hd[8]; float fl[3]; float r2c[16], c2w[16];
    rd(f, hd, 8); rd(f, fl, 3); rd(f, r2c, 16); rd(f, c2w, 16);
    const int xres = hd[0], yres = hd[1], spp = hd[2], maxDepth = hd[3], strat = hd[4], nV = hd[5], nT = hd[6], nM = hd[7];
    std::vector<float> verts((size_t)nV * 3), Le((size_t)nT * 3);
    std::vector<int32_t> idx((size_t)nT * 3), mat(nT), emis(nT);
    std::vector<EnigmaMaterial> mats(nM);
    rd(f, verts.data(), verts.size()); rd(f, idx.data(), idx.size()); rd(f, mat.data(), mat.size()); rd(f, emis.data(), emis.size());
    rd(f, Le.data(), Le.size()); rd(f, mats.data(), mats.size());
    if (!f) { std::cerr << "truncated scene file\n"; return 2; }

    // film + camera + sampler (the reference's defaults: box filter 0.5, full crop window)
    std::unique_ptr<Filter> filter(new BoxFilter(Vector2f(0.5f, 0.5f)));
    std::unique_ptr<Film> film(new Film(Vector2i(xres, yres), BBox2f(), std::move(filter), 35.f, "", 1.f));
    Film* filmPtr = film.get();
    Transform T_r2c, T_c2w;
    memcpy(T_r2c.m_mat, r2c, 64); memcpy(T_c2w.m_mat, c2w, 64);
    CameraPtr camera(new PerspectiveCamera(T_c2w, T_r2c, 0.f, 1.f, fl[1], fl[2], std::move(film)));
    SamplerPtr sampler(new SobolSampler(spp, filmPtr->GetSampleBounds()));

    // scene: one mesh per triangle run sharing material / emission (keeps the file's triangle order)
    Scene scene;
    for (const EnigmaMaterial& m : mats) scene.AddMaterial(m);
    for (int t = 0; t < nT;) {
        int e = t;
        while (e < nT && mat[e] == mat[t] && emis[e] == emis[t] && memcmp(&Le[3 * e], &Le[3 * t], 12) == 0) ++e;
        TriangleMesh m;
        m.m_nTriangles = e - t; m.materialIndex = mat[t]; m.emissive = emis[t] != 0; m.Le = Spectrum(Le[3 * t], Le[3 * t + 1], Le[3 * t + 2]);
        for (int k = t; k < e; ++k)
            for (int v = 0; v < 3; ++v) {
                const float* p = &verts[3 * (size_t)idx[3 * k + v]];
                m.m_p.push_back(Vector3f(p[0], p[1], p[2]));
                m.m_vertexIndices.push_back((int)m.m_p.size() - 1);
            }
        m.m_nVertices = (int)m.m_p.size();
        scene.AddTriangleMesh(m);
        t = e;
    }
    // integrator (CreatePathIntegrator: pixelBounds = film sample bounds, Enigma/PathIntegrator.cpp:187)
    // (CreateDirectLightingIntegrator: the same pixel bounds, Enigma/DirectLightingIntegrator.cpp:104)
    std::unique_ptr<GpuSamplerIntegrator> integrator;
    if (strat >= 2)
        integrator.reset(new GpuDirectLightingIntegrator(strat == 2 ? GpuDirectLightingIntegrator::eLightStrategy::UniformSampleAll
                                                                    : GpuDirectLightingIntegrator::eLightStrategy::UniformSampleOne,
                                                         maxDepth, camera, sampler, filmPtr->GetSampleBounds()));
    else integrator.reset(new GpuPathIntegrator(maxDepth, camera, sampler, filmPtr->GetSampleBounds(), fl[0], strat ? "power" : "uniform"));
    try {
        integrator->Render(scene);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    const std::vector<float>& rgb = filmPtr->GetRGB();
    std::ofstream o(argv[2], std::ios::binary);
    o.write(reinterpret_cast<const char*>(rgb.data()), rgb.size() * sizeof(float));
    const EnigmaStats& st = integrator->GetStats();
    std::cout << "rendered " << xres << "x" << yres << " spp " << spp << " rays " << (st.rays_closest + st.rays_shadow + st.rays_mis) << " in " << st.ms_render << " ms\n";
    return 0;
}
