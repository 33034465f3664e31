// COLMAP-binary model I/O in the layout the reference reads and writes:
//   cameras.bin   u64 n; { u32 id, u32 model, u64 w, u64 h, f64 params[k] }           io_ecim.cc:9-29 / 145-159
//   images.bin    u64 n; { u32 id, f64 qw qx qy qz, f64 t[3], u32 camera, cstring name,
//                          u64 n2d, { f64 x, f64 y, u64 track (all ones = none) } }    io_ecim.cc:31-57 / 161-191
//   points3D.bin  u64 n; { u64 id, f64 xyz, u8 rgb[3], f64 error, u64 nobs, { i32 frame, i32 p2d } }   io_ecim.cc:59-84 / 193-222
// (/root/reference/src/utility/io_ecim.cc).  Parameter counts per model: src/base/camera_model.hpp (3,4,4,5,8).
// Own implementation on plain structs; used by tools/ba_replay.cc to run BA-only replays of stored reconstructions.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace xrsfm_amd {

struct ModelCamera { uint32_t id = 0, model = 0; uint64_t w = 0, h = 0; std::vector<double> params; };
struct ModelPoint2D { double x = 0, y = 0; uint64_t track = ~0ull; };
struct ModelImage { uint32_t id = 0, camera = 0; double q[4] = {1, 0, 0, 0} /* w x y z */, t[3] = {0, 0, 0}; std::string name; std::vector<ModelPoint2D> points; };
struct ModelPoint3D { uint64_t id = 0; double xyz[3] = {0, 0, 0}; uint8_t rgb[3] = {0, 0, 0}; double error = -1; std::vector<std::pair<int32_t, int32_t>> obs; };
struct Model { std::vector<ModelCamera> cameras; std::vector<ModelImage> images; std::vector<ModelPoint3D> points; };

inline int model_num_params(uint32_t model) { static const int n[5] = {3, 4, 4, 5, 8}; return model < 5 ? n[model] : -1; }

namespace detail {
template <typename T> inline bool rd(FILE* f, T* v, size_t n = 1) { return fread(v, sizeof(T), n, f) == n; }
template <typename T> inline bool wr(FILE* f, const T* v, size_t n = 1) { return fwrite(v, sizeof(T), n, f) == n; }
struct File { FILE* f; File(const std::string& p, const char* m) : f(fopen(p.c_str(), m)) {} ~File() { if (f) fclose(f); } };
}  // namespace detail

inline bool read_model(const std::string& dir, Model& m) {
    using namespace detail;
    {
        File fh(dir + "/cameras.bin", "rb");
        uint64_t n = 0;
        if (!fh.f || !rd(fh.f, &n)) return false;
        m.cameras.resize(n);
        for (auto& c : m.cameras) {
            if (!rd(fh.f, &c.id) || !rd(fh.f, &c.model) || !rd(fh.f, &c.w) || !rd(fh.f, &c.h)) return false;
            const int k = model_num_params(c.model);
            if (k < 0) return false;
            c.params.resize(k);
            if (!rd(fh.f, c.params.data(), k)) return false;
        }
    }
    {
        File fh(dir + "/images.bin", "rb");
        uint64_t n = 0;
        if (!fh.f || !rd(fh.f, &n)) return false;
        m.images.resize(n);
        for (auto& im : m.images) {
            if (!rd(fh.f, &im.id) || !rd(fh.f, im.q, 4) || !rd(fh.f, im.t, 3) || !rd(fh.f, &im.camera)) return false;
            im.name.clear();
            for (;;) { char ch; if (!rd(fh.f, &ch)) return false; if (ch == '\0') break; im.name += ch; }
            uint64_t n2 = 0;
            if (!rd(fh.f, &n2)) return false;
            im.points.resize(n2);
            for (auto& p : im.points) if (!rd(fh.f, &p.x) || !rd(fh.f, &p.y) || !rd(fh.f, &p.track)) return false;
        }
    }
    {
        File fh(dir + "/points3D.bin", "rb");
        uint64_t n = 0;
        if (!fh.f || !rd(fh.f, &n)) return false;
        m.points.resize(n);
        for (auto& p : m.points) {
            uint64_t no = 0;
            if (!rd(fh.f, &p.id) || !rd(fh.f, p.xyz, 3) || !rd(fh.f, p.rgb, 3) || !rd(fh.f, &p.error) || !rd(fh.f, &no)) return false;
            p.obs.resize(no);
            for (auto& o : p.obs) if (!rd(fh.f, &o.first) || !rd(fh.f, &o.second)) return false;
        }
    }
    return true;
}

inline bool write_model(const std::string& dir, const Model& m) {
    using namespace detail;
    {
        File fh(dir + "/cameras.bin", "wb");
        const uint64_t n = m.cameras.size();
        if (!fh.f || !wr(fh.f, &n)) return false;
        for (const auto& c : m.cameras)
            if (!wr(fh.f, &c.id) || !wr(fh.f, &c.model) || !wr(fh.f, &c.w) || !wr(fh.f, &c.h) || !wr(fh.f, c.params.data(), c.params.size())) return false;
    }
    {
        File fh(dir + "/images.bin", "wb");
        const uint64_t n = m.images.size();
        if (!fh.f || !wr(fh.f, &n)) return false;
        for (const auto& im : m.images) {
            const uint64_t n2 = im.points.size();
            if (!wr(fh.f, &im.id) || !wr(fh.f, im.q, 4) || !wr(fh.f, im.t, 3) || !wr(fh.f, &im.camera)) return false;
            if (!wr(fh.f, im.name.c_str(), im.name.size() + 1) || !wr(fh.f, &n2)) return false;
            for (const auto& p : im.points) if (!wr(fh.f, &p.x) || !wr(fh.f, &p.y) || !wr(fh.f, &p.track)) return false;
        }
    }
    {
        File fh(dir + "/points3D.bin", "wb");
        const uint64_t n = m.points.size();
        if (!fh.f || !wr(fh.f, &n)) return false;
        for (const auto& p : m.points) {
            const uint64_t no = p.obs.size();
            if (!wr(fh.f, &p.id) || !wr(fh.f, p.xyz, 3) || !wr(fh.f, p.rgb, 3) || !wr(fh.f, &p.error) || !wr(fh.f, &no)) return false;
            for (const auto& o : p.obs) if (!wr(fh.f, &o.first) || !wr(fh.f, &o.second)) return false;
        }
    }
    return true;
}

}  // namespace xrsfm_amd
