// Host driver written against the mirrored reference interface (enigma_host.h): loads a flat scene
// file, builds Film / PerspectiveCamera / SobolSampler / Scene / GpuPathIntegrator (or GpuDirectLightingIntegrator) the way
// RenderOptions::MakeIntegrator + MakeScene do (Enigma/Api.cpp:1170-1254), calls
// integrator->Render(scene) like Enigma/Main.h:245, and writes the film's RGB as raw floats.
//
//   enigma_render <scene.bin> <out.f32>
//   enigma_render <scene.pbrt> [out.pfm | out.f32]      scene description through the pbrt front end (pbrt_frontend.h)
//   enigma_render --dump <scene.pbrt> <out.json>        parse only (no GPU): what the front end hands to the integrator
//   enigma_render --gpus N <scene> <out>                 one child process per GPU (ranks 0..N-1 on devices 0..N-1): each renders its
//                                                        share of the sample numbers, RCCL sums the films onto rank 0, which writes <out>
// scene.bin (little endian): "ESC1", then int32 {xres, yres, spp, maxDepth, integrator (PathIntegrator: 0 uniform / 1 power
//   light distribution; DirectLightingIntegrator: 2 strategy all / 3 strategy one),
//   nVerts, nTris, nMaterials}, float {rrThreshold, lensRadius, focalDistance}, float[16] rasterToCamera,
//   float[16] cameraToWorld, float[nVerts*3], int32[nTris*3], int32[nTris] material,
//   int32[nTris] emissive flag, float[nTris*3] Le, EnigmaMaterial[nMaterials].
#include "enigma_host.h"
#include "pbrt_frontend.h"

#include <algorithm>
#include <chrono>
#include <csignal>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

#include <sys/wait.h>
#include <unistd.h>

using namespace RayTrace;

template <class T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); }

static bool endsWith(const std::string& s, const char* suffix) {
    size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

static void dumpJson(const FrontEndResult& R, std::ostream& o) {
    o.precision(9);
    auto arr = [&](const float* v, size_t n) { o << "["; for (size_t i = 0; i < n; ++i) o << (i ? "," : "") << (std::isinf(v[i]) ? (v[i] > 0 ? 1e39 : -1e39) : (double)v[i]); o << "]"; };
    const Film& film = R.camera->GetFilm();
    const bool environment = dynamic_cast<const EnvironmentCamera*>(R.camera.get()) != nullptr;
    static const ProjectiveCamera noProjection(Transform(), Transform(), 0.f, 1.f, 0.f, 1e30f, nullptr);
    const ProjectiveCamera& cam = environment ? noProjection : static_cast<const ProjectiveCamera&>(*R.camera);
    const HaltonSampler* hal = dynamic_cast<const HaltonSampler*>(R.sampler.get());
    const Sampler& sob = *R.sampler;
    const BBox2i &cb = film.m_croppedPixelBounds, &sb = hal ? hal->m_sampleBounds : static_cast<const SobolSampler&>(*R.sampler).m_sampleBounds, &pb = R.pixelBounds;
    o << "{\"film\":{\"xres\":" << film.m_fullResolution.x << ",\"yres\":" << film.m_fullResolution.y << ",\"cropped\":[" << cb.m_min.x << "," << cb.m_min.y << ","
      << cb.m_max.x << "," << cb.m_max.y << "],\"filter\":\"" << R.filterName << "\",\"radius\":[" << film.m_filter->m_radius.x << "," << film.m_filter->m_radius.y
      << "],\"filename\":\"" << film.m_filename << "\",\"max_sample_luminance\":";
    float msl = film.GetMaxSampleLuminance(); arr(&msl, 1);
    o << ",\"filter_table\":"; arr(film.GetFilterTable(), 256);
    o << "},\n\"camera\":{\"type\":\"" << (environment ? "environment" : "perspective") << "\",\"c2w\":"; arr(&R.camera->m_cameraToWorld.m_mat[0][0], 16);
    o << ",\"r2c\":"; arr(&cam.m_rasterToCamera.m_mat[0][0], 16);
    o << ",\"lensradius\":"; arr(&cam.m_lensRadius, 1); o << ",\"focaldistance\":"; arr(&cam.m_focalDistance, 1);
    o << "},\n\"sampler\":{\"name\":\"" << (hal ? "halton" : "sobol") << "\",\"samplepixelcenter\":" << (hal && hal->m_sampleAtPixelCenter ? "true" : "false")
      << ",\"spp\":" << sob.m_samplesPerPixel << ",\"sample_bounds\":[" << sb.m_min.x << "," << sb.m_min.y << "," << sb.m_max.x << "," << sb.m_max.y << "]},\n";
    o << "\"integrator\":{\"name\":\"" << R.integratorName << "\",\"maxdepth\":" << R.maxDepth << ",\"rrthreshold\":" << R.rrThreshold << ",\"lightsamplestrategy\":\""
      << R.lightStrategy << "\",\"strategy\":" << R.directStrategy << ",\"pixel_bounds\":[" << pb.m_min.x << "," << pb.m_min.y << "," << pb.m_max.x << "," << pb.m_max.y << "]},\n";
    o << "\"materials\":[";
    for (size_t i = 0; i < R.scene.m_materials.size(); ++i) {
        const EnigmaMaterial& m = R.scene.m_materials[i];
        o << (i ? "," : "") << "{\"type\":" << m.type << ",\"remap\":" << m.remap_roughness << ",\"kd\":"; arr(m.kd, 3);
        o << ",\"ks\":"; arr(m.ks, 3); o << ",\"kr\":"; arr(m.kr, 3); o << ",\"kt\":"; arr(m.kt, 3); o << ",\"eta\":"; arr(m.eta, 3); o << ",\"k\":"; arr(m.k, 3);
        o << ",\"sigma\":" << m.sigma << ",\"uroughness\":" << m.uroughness << ",\"vroughness\":" << m.vroughness << ",\"index\":" << m.index
          << ",\"tex\":[" << m.tex_kd << "," << m.tex_ks << "," << m.tex_kr << "," << m.tex_kt << "," << m.tex_eta << "," << m.tex_k << "]";
        o << ",\"reflect\":"; arr(m.reflect, 3); o << ",\"transmit\":"; arr(m.transmit, 3); o << ",\"opacity\":"; arr(m.opacity, 3); o << ",\"amount\":"; arr(m.amount, 3);
        o << ",\"mix\":[" << m.mix_a << "," << m.mix_b << "]}";
    }
    o << "],\n\"meshes\":[";
    for (size_t i = 0; i < R.scene.m_meshes.size(); ++i) {
        const TriangleMesh& m = R.scene.m_meshes[i];
        o << (i ? ",\n" : "") << "{\"material\":" << m.materialIndex << ",\"medium_inside\":" << m.mediumInside << ",\"medium_outside\":" << m.mediumOutside << ",\"reverse\":" << ((m.reverseOrientation ^ m.transformSwapsHandedness) ? 1 : 0)
          << ",\"reverse_orientation\":" << (m.reverseOrientation ? 1 : 0) << ",\"emissive\":" << (m.emissive ? 1 : 0)
          << ",\"twosided\":" << (m.twoSided ? 1 : 0) << ",\"nsamples\":" << m.nSamples << ",\"Le\":[" << m.Le[0] << "," << m.Le[1] << "," << m.Le[2] << "],\"indices\":[";
        for (size_t k = 0; k < m.m_vertexIndices.size(); ++k) o << (k ? "," : "") << m.m_vertexIndices[k];
        o << "],\"P\":[";
        for (size_t k = 0; k < m.m_p.size(); ++k) o << (k ? "," : "") << m.m_p[k].x << "," << m.m_p[k].y << "," << m.m_p[k].z;
        o << "],\"N\":[";
        for (size_t k = 0; k < m.m_n.size(); ++k) o << (k ? "," : "") << m.m_n[k].x << "," << m.m_n[k].y << "," << m.m_n[k].z;
        o << "],\"uv\":[";
        for (size_t k = 0; k < m.m_uv.size(); ++k) o << (k ? "," : "") << m.m_uv[k].x << "," << m.m_uv[k].y;
        o << "]}";
    }
    o << "],\n\"spheres\":[";
    for (size_t i = 0; i < R.scene.m_spheres.size(); ++i) {
        const EnigmaSphere& s = R.scene.m_spheres[i];
        o << (i ? "," : "") << "{\"o2w\":"; arr(s.object_to_world, 16); o << ",\"w2o\":"; arr(s.world_to_object, 16);
        o << ",\"radius\":" << s.radius << ",\"zmin\":" << s.z_min << ",\"zmax\":" << s.z_max << ",\"phimax\":" << s.phi_max << ",\"material\":" << s.material
          << ",\"flags\":" << s.flags << ",\"medium_inside\":" << s.medium_inside - 1 << ",\"medium_outside\":" << s.medium_outside - 1 << "}";
    }
    o << "],\n\"textures\":[";
    for (size_t i = 0; i < R.scene.m_textures.size(); ++i) {
        const EnigmaTexture& t = R.scene.m_textures[i];
        const std::vector<float>& px = R.scene.m_textureTexels[i];
        double sum = 0;
        for (float v : px) sum += v;
        o << (i ? "," : "") << "{\"width\":" << t.width << ",\"height\":" << t.height << ",\"trilinear\":" << t.trilinear << ",\"gamma\":" << t.gamma << ",\"scale\":" << t.scale
          << ",\"wrap\":" << t.wrap << ",\"mapping\":[" << t.su << "," << t.sv << "," << t.du << "," << t.dv << "],\"first_texel\":[" << px[0] << "," << px[1] << "," << px[2]
          << "],\"sum\":" << sum << "}";
    }
    o << "],\n\"media\":[";
    for (size_t i = 0; i < R.scene.m_media.size(); ++i) {
        const EnigmaMedium& m = R.scene.m_media[i];
        o << (i ? "," : "") << "{\"sigma_a\":"; arr(m.sigma_a, 3); o << ",\"sigma_s\":"; arr(m.sigma_s, 3); o << ",\"g\":"; arr(&m.g, 1); o << ",\"kind\":" << m.kind << "}";
    }
    o << "],\n\"lights\":[";
    for (size_t i = 0; i < R.scene.m_lights.size(); ++i) {
        const EnigmaLight& l = R.scene.m_lights[i].light;
        o << (i ? "," : "") << "{\"order\":" << R.scene.m_lights[i].order << ",\"type\":" << l.type << ",\"from\":"; arr(l.pos, 3);
        o << ",\"I\":"; arr(l.L, 3); o << ",\"dir\":"; arr(l.dir, 3);
        o << ",\"cos_total_width\":"; arr(&l.cos_total_width, 1); o << ",\"cos_falloff_start\":"; arr(&l.cos_falloff_start, 1);
        o << ",\"world_to_light\":"; arr(l.world_to_light, 9); o << ",\"light_to_world\":"; arr(l.light_to_world, 9); o << ",\"nsamples\":" << l.n_samples
          << ",\"map\":[" << l.map_width << "," << l.map_height << "]"
          << ",\"sphere\":" << (l.type == ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE ? (int)l.triangle : -1) << ",\"twosided\":" << l.two_sided << "}";
    }
    o << "],\n\"warnings\":[";
    for (size_t i = 0; i < R.warnings.size(); ++i) {
        std::string w = R.warnings[i];
        for (char& ch : w) if (ch == '"' || ch == '\\') ch = '\'';
        o << (i ? "," : "") << "\"" << w << "\"";
    }
    o << "]}\n";
}

// ---- one process per GPU ----
// The parent forks the ranks before anything touches the GPU; rank 0 asks the library for the communicator id and leaves
// it in a file beside the output, the other ranks wait for that file.
struct Ranks { int rank = 0, world = 1; unsigned char id[ENIGMA_COMM_ID_BYTES] = {}; };
static Ranks g_ranks;

static bool exchangeCommId(const std::string& path, Ranks* r) {
    if (r->rank == 0) {
        if (enigma_comm_unique_id(r->id) != ENIGMA_OK) { std::cerr << "enigma_comm_unique_id failed (RCCL not loadable, or no GPU visible to it)\n"; return false; }
        const std::string tmp = path + ".part";
        { std::ofstream o(tmp, std::ios::binary); o.write(reinterpret_cast<const char*>(r->id), sizeof(r->id)); if (!o) return false; }
        return rename(tmp.c_str(), path.c_str()) == 0; // appears whole or not at all
    }
    for (int tries = 0; tries < 6000; ++tries) { // up to a minute
        std::ifstream f(path, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(r->id), sizeof(r->id))) return true;
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    std::cerr << "rank " << r->rank << ": no communicator id from rank 0\n";
    return false;
}

// returns -1 in a child (which goes on to render as g_ranks.rank), else the parent's exit code.
// The film reduce and the communicator set-up are collectives without a timeout: a rank that dies (no such GPU, a scene it
// cannot load, an error of the library) would leave its peers waiting inside RCCL for good.  The parent therefore watches
// all ranks at once: the first one that ends badly -- or a deadline, ENIGMA_RENDER_DEADLINE_S seconds, default none -- ends
// the others (SIGKILL: a rank blocked inside a collective handles nothing else), every child is reaped, and the exit code
// says that the job failed.
static int forkRanks(int world, const std::string& out) {
    const std::string idFile = out + ".commid";
    unlink(idFile.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        pid_t pid = fork();
        if (pid < 0) {
            std::cerr << "fork failed\n";
            for (pid_t k : kids) kill(k, SIGKILL);
            for (pid_t k : kids) waitpid(k, nullptr, 0);
            return 1;
        }
        if (pid == 0) {
            g_ranks.rank = r; g_ranks.world = world;
            if (!exchangeCommId(idFile, &g_ranks)) _exit(3);
            return -1;
        }
        kids.push_back(pid);
    }
    double deadline = 0;
    if (const char* e = getenv("ENIGMA_RENDER_DEADLINE_S")) deadline = atof(e);
    const auto t0 = std::chrono::steady_clock::now();
    size_t left = kids.size();
    int rc = 0;
    while (left > 0) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, WNOHANG);
        if (done > 0) {
            --left;
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                if (rc == 0) {
                    const size_t rank = std::find(kids.begin(), kids.end(), done) - kids.begin();
                    std::cerr << "enigma_render: rank " << rank << " failed (" << (WIFEXITED(st) ? "exit code " : "signal ") << (WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st))
                              << "); ending the other ranks\n";
                    for (pid_t k : kids) if (k != done) kill(k, SIGKILL); // already reaped ones are gone: ESRCH is harmless
                }
                rc = 1;
            }
            continue;
        }
        if (done < 0) { rc = 1; break; } // no children left to wait for although some were expected
        if (deadline > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline) {
            if (rc == 0) std::cerr << "enigma_render: the ranks did not finish within " << deadline << " s; ending them\n";
            rc = 1;
            deadline = 0;
            for (pid_t k : kids) kill(k, SIGKILL);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    unlink(idFile.c_str());
    return rc;
}

// Scene description file -> front end -> GPU integrator, like pbrtParseFile + pbrtWorldEnd in Enigma/Main.h.
static int runPbrt(const std::string& scene, const std::string& out, bool dumpOnly) {
    try {
        PbrtApi api(!dumpOnly, g_ranks.rank);
        if (g_ranks.world > 1) api.SetRanks(g_ranks.rank, g_ranks.world, g_ranks.id);
        if (!dumpOnly && endsWith(out, ".pfm")) api.SetImageFile(out);
        api.pbrtParseFile(scene);
        FrontEndResult& R = api.Result();
        if (!R.camera) { std::cerr << "scene description has no WorldEnd\n"; return 1; }
        for (const std::string& w : R.warnings) std::cerr << "Warning: " << w << "\n";
        if (dumpOnly) {
            std::ofstream o(out);
            dumpJson(R, o);
            return o ? 0 : 1;
        }
        if (g_ranks.rank != 0) return 0; // the film is rank 0's
        if (!out.empty() && !endsWith(out, ".pfm")) {
            const std::vector<float>& rgb = R.camera->m_film->GetRGB();
            std::ofstream o(out, std::ios::binary);
            o.write(reinterpret_cast<const char*>(rgb.data()), rgb.size() * sizeof(float));
        }
        const EnigmaStats& st = R.integrator->GetStats();
        std::cout << "rendered " << scene << " rays " << (st.rays_closest + st.rays_shadow + st.rays_mis) << " in " << st.ms_render << " ms\n";
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[1]) == "--dump") return runPbrt(argv[2], argv[3], true);
    if (argc >= 5 && std::string(argv[1]) == "--gpus") {
        const int world = atoi(argv[2]);
        if (world < 1 || world > 64) { std::cerr << "--gpus takes 1..64\n"; return 2; }
        argv += 2; argc -= 2;
        if (world > 1) {
            int rc = forkRanks(world, argv[2]);
            if (rc >= 0) return rc; // the parent; children fall through with g_ranks set
        }
    }
    if (argc >= 2 && endsWith(argv[1], ".pbrt")) return runPbrt(argv[1], argc >= 3 ? argv[2] : "", false);
    if (argc < 3) { std::cerr << "usage: enigma_render scene.bin out.f32 | scene.pbrt [out.pfm|out.f32] | --dump scene.pbrt out.json\n"; return 2; }
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
                                                         maxDepth, camera, sampler, filmPtr->GetSampleBounds(), g_ranks.rank));
    else integrator.reset(new GpuPathIntegrator(maxDepth, camera, sampler, filmPtr->GetSampleBounds(), fl[0], strat ? "power" : "uniform", g_ranks.rank));
    if (g_ranks.world > 1) integrator->SetRanks(g_ranks.rank, g_ranks.world, g_ranks.id);
    try {
        integrator->Render(scene);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    if (g_ranks.rank != 0) return 0; // the film is rank 0's
    const std::vector<float>& rgb = filmPtr->GetRGB();
    std::ofstream o(argv[2], std::ios::binary);
    o.write(reinterpret_cast<const char*>(rgb.data()), rgb.size() * sizeof(float));
    const EnigmaStats& st = integrator->GetStats();
    std::cout << "rendered " << xres << "x" << yres << " spp " << spp << " rays " << (st.rays_closest + st.rays_shadow + st.rays_mis) << " in " << st.ms_render << " ms\n";
    return 0;
}
