// BA-only replay of a stored reconstruction: read a COLMAP-binary model as the reference writes it
// (WriteColMapDataBinary, io_ecim.cc:224-235), run the global BA of BASolver::GBA on the MI355X through the C-ABI, write
// the refined model.  usage: ba_replay <model_dir_in> <model_dir_out> [--fast] [--filter max_re deg]
//   gauge: the translations of the first two images are held constant (the role of map.init_id1/2, ba_solver.cc:611-614)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/xrsfm_ba.h"
#include "../xrsfm_amd/csrc/io/colmap_model.h"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ba_replay <in_dir> <out_dir> [--fast] [--filter max_re deg]\n"); return 2; }
    bool fast = false, filter = false; double max_re = 4.0, deg = 1.5;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--fast")) fast = true;
        else if (!strcmp(argv[i], "--filter") && i + 2 < argc) { filter = true; max_re = atof(argv[i + 1]); deg = atof(argv[i + 2]); i += 2; }
    }
    xrsfm_amd::Model m;
    if (!xrsfm_amd::read_model(argv[1], m)) { fprintf(stderr, "cannot read model in %s\n", argv[1]); return 1; }
    std::map<uint32_t, int> cam_slot; std::map<uint64_t, int> pt_slot; std::map<uint32_t, int> img_slot;
    std::vector<int32_t> intr_model, cam_intr, obs_cam, obs_pt;
    std::vector<double> intr_params, cam_q, cam_t, points, obs_uv;
    std::vector<uint8_t> cam_const;
    for (const auto& c : m.cameras) {
        cam_slot[c.id] = (int)intr_model.size();
        intr_model.push_back((int32_t)c.model);
        for (int k = 0; k < 8; ++k) intr_params.push_back(k < (int)c.params.size() ? c.params[k] : 0.0);
    }
    for (const auto& p : m.points) { pt_slot[p.id] = (int)pt_slot.size(); points.insert(points.end(), p.xyz, p.xyz + 3); }
    for (const auto& im : m.images) {
        const int ci = (int)img_slot.size();
        img_slot[im.id] = ci;
        cam_q.insert(cam_q.end(), {im.q[1], im.q[2], im.q[3], im.q[0]});      // file: w x y z -> Eigen coeffs x y z w
        cam_t.insert(cam_t.end(), im.t, im.t + 3);
        cam_const.push_back(ci < 2 ? XRSFM_BA_CONST_T : 0);
        auto cs = cam_slot.find(im.camera);
        if (cs == cam_slot.end()) { fprintf(stderr, "image %u references unknown camera %u\n", im.id, im.camera); return 1; }
        cam_intr.push_back(cs->second);
        for (const auto& p2 : im.points) {
            if (p2.track == ~0ull) continue;
            auto ps = pt_slot.find(p2.track);
            if (ps == pt_slot.end()) continue;           // observation of a filtered track
            obs_cam.push_back(ci); obs_pt.push_back(ps->second); obs_uv.push_back(p2.x); obs_uv.push_back(p2.y);
        }
    }
    xrsfm_ba_problem p;
    p.n_cams = (int32_t)cam_intr.size(); p.n_points = (int32_t)(points.size() / 3); p.n_obs = (int32_t)obs_cam.size(); p.n_intr = (int32_t)intr_model.size();
    p.cam_q = cam_q.data(); p.cam_t = cam_t.data(); p.cam_const = cam_const.data(); p.cam_intr = cam_intr.data();
    p.intr_model = intr_model.data(); p.intr_params = intr_params.data(); p.points = points.data(); p.point_const = nullptr;
    p.obs_cam = obs_cam.data(); p.obs_pt = obs_pt.data(); p.obs_uv = obs_uv.data();
    xrsfm_ba_options opt; xrsfm_ba_default_options(&opt);
    if (fast) { opt.max_iterations = 20; opt.function_tolerance = 1e-4; opt.parameter_tolerance = 1e-5; }
    opt.verbose = 1;
    xrsfm_ba_summary s;
    const int rc = xrsfm_ba_solve(&opt, &p, &s);
    if (rc != XRSFM_BA_OK) { fprintf(stderr, "xrsfm_ba_solve failed: %d\n", rc); return 1; }
    const double nres = s.num_residuals > 0 ? s.num_residuals : 1;
    printf("cameras %d points %d observations %d | iterations %d | cost %.6f -> %.6f px | %.3f s\n", p.n_cams, p.n_points, p.n_obs,
           s.n_successful + s.n_unsuccessful, std::sqrt(s.initial_cost / nres), std::sqrt(s.final_cost / nres), s.total_time_s);
    std::vector<uint8_t> obs_del(p.n_obs, 0), trk_out(p.n_points, 0);
    std::vector<double> trk_err(p.n_points, -1.0);
    if (filter) {
        int32_t cnt[2] = {0, 0};
        const int rf = xrsfm_ba_filter_tracks(&p, max_re, deg * 3.14159265358979323846 / 180.0, obs_del.data(), trk_out.data(), trk_err.data(), nullptr, cnt);
        if (rf != XRSFM_BA_OK) { fprintf(stderr, "xrsfm_ba_filter_tracks failed: %d\n", rf); return 1; }
        printf("Outlier num1: %d Outlier num2: %d\n", cnt[0], cnt[1]);
    }
    // write back
    for (auto& im : m.images) {
        const int ci = img_slot[im.id];
        im.q[0] = cam_q[4 * ci + 3]; im.q[1] = cam_q[4 * ci]; im.q[2] = cam_q[4 * ci + 1]; im.q[3] = cam_q[4 * ci + 2];
        for (int k = 0; k < 3; ++k) im.t[k] = cam_t[3 * ci + k];
    }
    xrsfm_amd::Model out;
    out.cameras = m.cameras; out.images = m.images;
    for (auto& p3 : m.points) {
        const int j = pt_slot[p3.id];
        if (trk_out[j]) continue;
        for (int k = 0; k < 3; ++k) p3.xyz[k] = points[3 * j + k];
        if (filter) p3.error = trk_err[j];
        out.points.push_back(p3);
    }
    if (filter) {   // drop the filtered observations / tracks from the images and the tracks
        std::map<std::pair<int, int>, uint8_t> del;     // (image slot, point slot) -> deleted
        for (int i = 0; i < p.n_obs; ++i) if (obs_del[i] || trk_out[obs_pt[i]]) del[{obs_cam[i], obs_pt[i]}] = 1;
        for (auto& im : out.images)
            for (auto& p2 : im.points) {
                if (p2.track == ~0ull) continue;
                auto ps = pt_slot.find(p2.track);
                if (ps == pt_slot.end() || del.count({img_slot[im.id], ps->second})) p2.track = ~0ull;
            }
        for (auto& p3 : out.points) {
            std::vector<std::pair<int32_t, int32_t>> keep;
            for (const auto& o : p3.obs) { auto is = img_slot.find((uint32_t)o.first); if (is != img_slot.end() && !del.count({is->second, pt_slot[p3.id]})) keep.push_back(o); }
            p3.obs.swap(keep);
        }
    }
    if (!xrsfm_amd::write_model(argv[2], out)) { fprintf(stderr, "cannot write model to %s\n", argv[2]); return 1; }
    return 0;
}
