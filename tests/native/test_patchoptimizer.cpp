// A caller written the way the reference's own callers are (Scene::initPatches, src/hpmvs/Scene.cpp:94-178;
// CellProcessor::extend, src/hpmvs/CellProcessor.cpp:129): build the scene, construct
// mo3d::PatchOptimizer(options, &scene), call optimize(patch) and read the patch back.
// Input: a binary dump written by tests/test_gpu_cpp_interface.py.  Output: one text line per patch.
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <cmath>
#include <vector>

#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/PatchOptimizer.h>
#include <hpmvs/Scene.h>

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) { perror("read"); exit(2); } return v; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
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
    // what Scene::addCameras does, with the pixels handed over instead of decoded from JPEG files
    for (int i = 0; i < nv; i++) {
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
    std::vector<mo3d::Ppatch3d> patches;
    for (int i = 0; i < np; i++) {
        mo3d::Ppatch3d p(new mo3d::Patch3d);
        for (int k = 0; k < 4; k++) p->center_[k] = rd<float>(f);
        for (int k = 0; k < 4; k++) p->normal_[k] = rd<float>(f);
        p->scale_3dx_ = rd<float>(f);
        const int m = rd<int>(f);
        for (int k = 0; k < m; k++) p->images_.push_back(rd<int>(f));
        patches.push_back(p);
    }
    fclose(f);

    // one optimizer per "thread", held by value like the reference does
    std::vector<mo3d::PatchOptimizer> optimizers;
    for (int i = 0; i < 2; i++) optimizers.emplace_back(options, &scene);

    // (1) the unchanged-caller form: one patch at a time on copies
    const int nsingle = np < 8 ? np : 8;
    for (int i = 0; i < nsingle; i++) {
        mo3d::Patch3d copy(*patches[i]);
        const bool ok = optimizers[i % 2].optimize(copy);
        printf("single %d %d %.9g %.9g %.9g %.9g %.9g %.9g %zu\n", i, ok ? 1 : 0, copy.center_[0], copy.center_[1],
               copy.center_[2], copy.normal_[0], copy.normal_[1], copy.normal_[2], copy.images_.size());
    }
    // (1b) the reference's data-parallel form (src/hpmvs/Scene.cpp:94-96,114,166-167): one optimizer per OpenMP
    //      thread, all calling optimize() concurrently against the one shared scene -- with 4 threads, and with
    //      40 (more concurrent launches than the scene has launch workspaces: they must queue, not collide)
    for (int nthreads : {4, 40}) {
        std::vector<mo3d::PatchOptimizer> pool;
        for (int i = 0; i < nthreads; i++) pool.emplace_back(options, &scene);
        std::vector<mo3d::Patch3d> copies;
        for (int i = 0; i < np; i++) copies.emplace_back(*patches[i]);
        std::vector<int> okv(np, 0);
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int i = 0; i < np; i++) okv[i] = pool[omp_get_thread_num()].optimize(copies[i]) ? 1 : 0;
        for (int i = 0; i < np; i++)
            printf("omp%d %d %d %.9g %.9g %.9g %.9g %.9g %.9g %zu\n", nthreads, i, okv[i], copies[i].center_[0], copies[i].center_[1],
                   copies[i].center_[2], copies[i].normal_[0], copies[i].normal_[1], copies[i].normal_[2],
                   copies[i].images_.size());
    }
    // (2) the additive batch entry
    std::vector<mo3d::Patch3d> originals;  // the seeds as read (the batch entry refines `patches` in place)
    for (auto& p : patches) originals.emplace_back(*p);
    std::vector<mo3d::Patch3d*> raw;
    for (auto& p : patches) raw.push_back(p.get());
    std::vector<unsigned char> ok(np);
    const size_t good = optimizers[0].optimizeBatch(raw.data(), raw.size(), ok.data());
    for (int i = 0; i < np; i++) {
        const mo3d::Patch3d& p = *patches[i];
        printf("batch %d %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %zu", i, (int)ok[i], p.center_[0], p.center_[1],
               p.center_[2], p.normal_[0], p.normal_[1], p.normal_[2], p.color_[0], p.color_[1], p.color_[2], p.ncc_,
               p.images_.size());
        for (int id : p.images_) printf(" %d", id);
        printf("\n");
    }
    // (3) frontier expansion the way CellProcessor::extend would batch it: every refined patch is a parent whose
    //     leaf is centred on it with width 2*scale/0.9
    {
        std::vector<const mo3d::Patch3d*> parents;
        std::vector<mo3d::PatchOptimizer::CellRef> cells;
        for (int i = 0; i < np; i++)
            if (ok[i]) {
                parents.push_back(patches[i].get());
                mo3d::PatchOptimizer::CellRef c;
                c.c = Eigen::Vector3f(patches[i]->center_[0], patches[i]->center_[1], patches[i]->center_[2]);
                c.width = patches[i]->scale_3dx_ * (2.0f / 0.9f);
                cells.push_back(c);
            }
        std::vector<mo3d::Patch3d> cand;
        std::vector<unsigned char> acc;
        const int N = optimizers[1].expandBatch(mo3d::PatchOptimizer::EXTEND, parents.data(), cells.data(), parents.size(),
                                                nullptr, cand, acc);
        for (size_t t = 0; t < cand.size(); t++)
            printf("expand %zu %d %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g %zu %d\n", t, N, (int)acc[t], cand[t].center_[0],
                   cand[t].center_[1], cand[t].center_[2], cand[t].normal_[0], cand[t].normal_[1], cand[t].normal_[2],
                   cand[t].scale_3dx_, cand[t].images_.size(), cand[t].expanded_ ? 1 : 0);
    }
    // (4) the depth maps and the acceptance counts of the expansion, batched (Scene.cpp:351-381, 518-644)
    {
        std::vector<const mo3d::Patch3d*> acc;
        for (int i = 0; i < np; i++) if (ok[i]) acc.push_back(patches[i].get());
        std::vector<int> vis, blk, fre, vis2, blk2, fre2;
        bool fine = scene.resetDepths() && scene.depthGates(acc.data(), acc.size(), 1.0f, vis, blk, fre);
        fine = fine && scene.setDepths(acc.data(), acc.size()) && scene.depthGates(acc.data(), acc.size(), 1.0f, vis2, blk2, fre2);
        for (size_t t = 0; t < acc.size() && fine; t++)
            printf("depth %zu %d %d %d %d %d %d\n", t, vis[t], blk[t], fre[t], vis2[t], blk2[t], fre2[t]);
        if (!fine) printf("depth failed\n");
    }
    // (6) one priority level of CellProcessor::extend as a batched frontier with the reference's sequential result
    //     (PatchOptimizer::extendLevel): the first 60 refined patches are the level's leaves, their depths are in the maps
    {
        std::vector<const mo3d::Patch3d*> leaves;
        for (int i = 0; i < np && leaves.size() < 60; i++) if (ok[i]) leaves.push_back(patches[i].get());
        if (!leaves.empty()) {
            const float width = leaves[0]->scale_3dx_ * (2.0f / 0.9f);
            std::unordered_set<uint64_t> occupied;
            auto key = [&](const mo3d::Patch3d& p) {
                const long long ix = (long long)std::floor(p.center_[0] / width), iy = (long long)std::floor(p.center_[1] / width), iz = (long long)std::floor(p.center_[2] / width);
                return (uint64_t)(((ix + (1 << 20)) << 42) | ((iy + (1 << 20)) << 21) | (iz + (1 << 20)));
            };
            for (const mo3d::Patch3d* p : leaves) occupied.insert(key(*p));
            mo3d::PatchOptimizer::LevelResult R;
            bool fine = scene.resetDepths() && scene.setDepths(leaves.data(), leaves.size());
            fine = fine && optimizers[1].extendLevel(leaves.data(), leaves.size(), width, occupied, 1.0f, false, R);
            if (!fine) printf("level failed\n");
            else {
                for (size_t t = 0; t < R.stage.size(); t++)
                    printf("level %zu %d %d %d %d\n", t, R.stage[t], R.counts[3 * t], R.counts[3 * t + 1], R.counts[3 * t + 2]);
                printf("levelsummary %d %zu %zu %.9g\n", R.waves, R.accepted.size(), occupied.size(), width);
            }
        }
    }
    // (7) one priority level of CellProcessor::branch (PatchOptimizer::branchLevel): the first 60 refined patches are the leaves'
    //     patches, every leaf centred on its patch (width 2 * scale / 0.9), every second one on the final tree level; the maps
    //     afterwards are fingerprinted by the three gate counts of the leaves' patches
    {
        std::vector<const mo3d::Patch3d*> leaves;
        for (int i = 0; i < np && leaves.size() < 60; i++) if (ok[i]) leaves.push_back(patches[i].get());
        if (!leaves.empty()) {
            std::vector<mo3d::PatchOptimizer::CellRef> cells(leaves.size());
            std::vector<unsigned char> fin(leaves.size());
            for (size_t i = 0; i < leaves.size(); i++) {
                cells[i].c = Eigen::Vector3f(leaves[i]->center_[0], leaves[i]->center_[1], leaves[i]->center_[2]);
                cells[i].width = leaves[i]->scale_3dx_ * (2.0f / 0.9f);
                fin[i] = (unsigned char)(i & 1);
            }
            mo3d::PatchOptimizer::BranchResult B;
            std::vector<int> vis, blk, fre;
            bool fine = scene.resetDepths() && scene.setDepths(leaves.data(), leaves.size());
            fine = fine && optimizers[1].branchLevel(leaves.data(), cells.data(), leaves.size(), fin.data(), B);
            fine = fine && scene.depthGates(leaves.data(), leaves.size(), 1.0f, vis, blk, fre);
            if (!fine) printf("branch failed\n");
            else
                for (size_t i = 0; i < leaves.size(); i++)
                    printf("branch %zu %d %d %d %d %d %d %d %d %d\n", i, B.support[i], (int)B.split[i], (int)B.child[4 * i], (int)B.child[4 * i + 1],
                           (int)B.child[4 * i + 2], (int)B.child[4 * i + 3], vis[i], blk[i], fre[i]);
        }
    }
    // (5) the per-thread cache of pinned blocks stays under its cap whatever sizes the rounds have (a frontier
    //     loop whose round size changes every round must not pile up one pinned block per size class)
    {
        const size_t cap = (size_t)8 << 20;
        mo3d::PatchOptimizer::setPinnedCacheCap(cap);
        size_t worst = 0, biggest = 0;   // (the block returned last stays even when it alone exceeds the cap: ADVICE r5)
        std::vector<mo3d::Patch3d> copies;
        for (int rep = 0; rep < 12; rep++) {
            const size_t m = 4100 + 3000 * (size_t)rep;  // above the pinning threshold, a different size class every round
            copies.assign(m, originals[0]);
            for (size_t i = 0; i < m; i++) copies[i] = originals[i % np];
            std::vector<mo3d::Patch3d*> ptrs(m);
            for (size_t i = 0; i < m; i++) ptrs[i] = &copies[i];
            std::vector<unsigned char> okm(m);
            optimizers[0].optimizeBatch(ptrs.data(), m, okm.data());
            const size_t held = mo3d::PatchOptimizer::pinnedCacheBytes();
            worst = held > worst ? held : worst;
            biggest = m * 64 * 4 * 5 / 4 + 4096;   // images[]: the round's largest block, rounded up to its size class
            int same = 1;
            for (size_t i = 0; i < m; i++) same &= (okm[i] == ok[i % np]);
            if (!same) { printf("pincache results differ in round %d\n", rep); break; }
        }
        printf("pincache cap %zu worst %zu largest_block %zu\n", cap, worst, biggest);
    }
    printf("good %zu\n", good);
    return 0;
}
