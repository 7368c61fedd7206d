// One priority level of CellProcessor::extend at PRODUCTION size through the C++ host layer (PatchOptimizer::extendLevel:
// one batched refinement, conflict-free waves of depthGates / walk / setDepths, the reference's sequential result --
// reference src/hpmvs/CellProcessor.cpp:84-178, main.cpp:146-181), timed, with every candidate's fate written out so that
// tools/extend_level_scale.py can hold it against the oracle's true sequential loop.
//   bench_extend_level <scene dump> <out file> [levels = 1] [absInt = 0]
// Input: the dump format of tests/test_gpu_cpp_interface.py (_dump_scene); its patches are the level's LEAVES (already
// refined, with their image lists), in the scheduler's order.  Output (binary): per level
//   int32 n_leaves, int32 waves, int32 n_accepted, float width, double seconds(extendLevel), double seconds(setup)
//   then per candidate (6 n_leaves): int32 stage, int32 counts[3], float center[3], float normal[3], float scale, int32 n_images
// The accepted candidates of a level are the leaves of the next one.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/PatchOptimizer.h>
#include <hpmvs/Scene.h>

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) { perror("read"); exit(2); } return v; }
template <typename T> static void wr(FILE* f, const T& v) { if (fwrite(&v, sizeof(T), 1, f) != 1) { perror("write"); exit(2); } }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <scene dump> <out> [levels] [absInt]\n", argv[0]); return 2; }
    const int levels = argc > 3 ? atoi(argv[3]) : 1;
    const bool absInt = argc > 4 ? atoi(argv[4]) != 0 : false;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    mo3d::HpmvsOptions options;
    mo3d::Scene scene;
    mo3d::NVM_Model model;
    const int nv = rd<int>(f);
    std::vector<std::vector<unsigned char> > pix(nv);
    std::vector<int> W(nv), H(nv);
    for (int i = 0; i < nv; i++) {
        mo3d::NVM_Camera cam;
        cam.filename = "view" + std::to_string(i);
        W[i] = rd<int>(f); H[i] = rd<int>(f);
        cam.f = rd<double>(f);
        for (int k = 0; k < 4; k++) cam.rq[k] = rd<double>(f);
        for (int k = 0; k < 3; k++) cam.c[k] = rd<double>(f);
        cam.r = 0.0;
        pix[i].resize((size_t)W[i] * H[i] * 3);
        if (fread(pix[i].data(), 1, pix[i].size(), f) != pix[i].size()) return 2;
        model.cameras.push_back(cam);
    }
    for (int i = 0; i < nv; i++) {   // what Scene::addCameras does, with the pixels handed over instead of decoded from JPEG files
        scene.cameras_.emplace_back();
        scene.images_.emplace_back();
        scene.dict_[model.cameras[i].filename] = i;
        scene.images_[i].init(&model.cameras[i], options.MAXLEVEL);
        scene.images_[i].setPixels(W[i], H[i], pix[i].data());
        scene.cameras_[i].init(&model.cameras[i], scene.images_[i].getWidth(), scene.images_[i].getHeight(), options.MAXLEVEL);
    }
    scene.covis_.resize(nv);
    for (int i = 0; i < nv; i++) {
        const int m = rd<int>(f);
        for (int k = 0; k < m; k++) scene.covis_[i].push_back(rd<int>(f));
    }
    const int np = rd<int>(f);
    std::vector<mo3d::Patch3d> leaves(np);
    for (int i = 0; i < np; i++) {
        mo3d::Patch3d& p = leaves[i];
        for (int k = 0; k < 4; k++) p.center_[k] = rd<float>(f);
        for (int k = 0; k < 4; k++) p.normal_[k] = rd<float>(f);
        p.scale_3dx_ = rd<float>(f);
        const int m = rd<int>(f);
        for (int k = 0; k < m; k++) p.images_.push_back(rd<int>(f));
    }
    const float width = rd<float>(f);   // the level's leaf width (the scheduler's)
    fclose(f);

    mo3d::PatchOptimizer optimizer(options, &scene);
    std::unordered_set<uint64_t> occupied;
    auto key = [&](const mo3d::Patch3d& p) {
        const long long ix = (long long)std::floor(p.center_[0] / width), iy = (long long)std::floor(p.center_[1] / width), iz = (long long)std::floor(p.center_[2] / width);
        return (uint64_t)(((ix + (1 << 20)) << 42) | ((iy + (1 << 20)) << 21) | (iz + (1 << 20)));
    };
    for (const mo3d::Patch3d& p : leaves) occupied.insert(key(p));
    FILE* g = fopen(argv[2], "wb");
    if (!g) return 2;
    {   // the state the seed round leaves: every leaf's depths in the maps (also warms the device scene up)
        std::vector<const mo3d::Patch3d*> lp(leaves.size());
        for (size_t i = 0; i < leaves.size(); i++) lp[i] = &leaves[i];
        if (!scene.resetDepths() || !scene.setDepths(lp.data(), lp.size())) { fprintf(stderr, "depth setup failed\n"); return 1; }
    }
    for (int level = 0; level < levels && !leaves.empty(); level++) {
        std::vector<const mo3d::Patch3d*> lp(leaves.size());
        for (size_t i = 0; i < leaves.size(); i++) lp[i] = &leaves[i];
        mo3d::PatchOptimizer::LevelResult R;
        const double t0 = now();
        const bool fine = optimizer.extendLevel(lp.data(), lp.size(), width, occupied, 1.0f, absInt, R);
        const double t1 = now();
        if (!fine) { fprintf(stderr, "extendLevel failed at level %d\n", level); return 1; }
        wr<int>(g, (int)leaves.size()); wr<int>(g, R.waves); wr<int>(g, (int)R.accepted.size()); wr<float>(g, width);
        wr<double>(g, t1 - t0); wr<double>(g, 0.0);
        for (size_t t = 0; t < R.stage.size(); t++) {
            const mo3d::Patch3d& q = R.candidates[t];
            wr<int>(g, R.stage[t]);
            for (int k = 0; k < 3; k++) wr<int>(g, R.counts[3 * t + k]);
            for (int k = 0; k < 3; k++) wr<float>(g, q.center_[k]);
            for (int k = 0; k < 3; k++) wr<float>(g, q.normal_[k]);
            wr<float>(g, q.scale_3dx_);
            wr<int>(g, (int)q.images_.size());
        }
        printf("level %d: %zu leaves, %zu candidates, %zu accepted, %d waves, %.4f s\n", level, leaves.size(), R.stage.size(),
               R.accepted.size(), R.waves, t1 - t0);
        std::vector<mo3d::Patch3d> next;
        for (size_t t : R.accepted) next.push_back(R.candidates[t]);
        leaves.swap(next);
    }
    fclose(g);
    return 0;
}
