// The reference's calling pattern under load: T host threads, one mo3d::PatchOptimizer each, every thread calling
// optimize(patch) in a loop over its share of the patches (src/hpmvs/Scene.cpp:94-96,114,166-167;
// CellProcessor.cpp:129,256) -- against ONE optimizeBatch() call over the same patches.  Checks that both give
// the same bytes, and prints throughput of both forms and the wall latency of a lone optimize() call.
//   bench_callers <scene dump of tests/test_gpu_cpp_interface.py> <threads>
#include <omp.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/PatchOptimizer.h>
#include <hpmvs/Scene.h>

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) { perror("read"); exit(2); } return v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool same(const mo3d::Patch3d& a, const mo3d::Patch3d& b) {
    return !memcmp(&a.center_[0], &b.center_[0], 16) && !memcmp(&a.normal_[0], &b.normal_[0], 16) && a.images_ == b.images_ &&
           !memcmp(&a.color_[0], &b.color_[0], 12) && a.scale_3dx_ == b.scale_3dx_;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int T = atoi(argv[2]);
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
    std::vector<mo3d::Patch3d> seeds(np);
    for (int i = 0; i < np; i++) {
        for (int k = 0; k < 4; k++) seeds[i].center_[k] = rd<float>(f);
        for (int k = 0; k < 4; k++) seeds[i].normal_[k] = rd<float>(f);
        seeds[i].scale_3dx_ = rd<float>(f);
        const int m = rd<int>(f);
        for (int k = 0; k < m; k++) seeds[i].images_.push_back(rd<int>(f));
    }
    fclose(f);

    mo3d::PatchOptimizer one(options, &scene);
    {   // warm-up: scene upload, kernel load
        mo3d::Patch3d w(seeds[0]);
        one.optimize(w);
    }
    // (1) one batch
    std::vector<mo3d::Patch3d> A(seeds);
    std::vector<mo3d::Patch3d*> raw;
    for (auto& p : A) raw.push_back(&p);
    std::vector<unsigned char> okA(np);
    double t0 = now();
    one.optimizeBatch(raw.data(), raw.size(), okA.data());
    const double t_batch = now() - t0;
    // (2) T threads, optimize() one patch at a time
    std::vector<mo3d::PatchOptimizer> pool;
    for (int i = 0; i < T; i++) pool.emplace_back(options, &scene);
    std::vector<mo3d::Patch3d> B(seeds);
    std::vector<unsigned char> okB(np);
    // (the caller's thread pool exists before the loop that is timed, as it does in a scheduler that runs round after round:
    //  creating 256 threads under the container's 16-CPU quota takes longer than the first dozen calls)
    int started = 0;
#pragma omp parallel num_threads(T) reduction(+ : started)
    started += 1;
    if (started != T) fprintf(stderr, "bench_callers: %d of %d threads\n", started, T);
    struct rusage ru0, ru1;
    getrusage(RUSAGE_SELF, &ru0);
    t0 = now();
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
    for (int i = 0; i < np; i++) okB[i] = pool[omp_get_thread_num()].optimize(B[i]) ? 1 : 0;
    const double t_calls = now() - t0;
    getrusage(RUSAGE_SELF, &ru1);
    const double cpu_user = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + 1e-6 * (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec);
    const double cpu_sys = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
    int mismatches = 0, good = 0;
    for (int i = 0; i < np; i++) {
        if (okA[i] != okB[i] || !same(A[i], B[i])) {
            mismatches++;
            fprintf(stderr, "bench_callers: patch %d differs: batch ok %d (%zu images, centre %.9g %.9g %.9g) / call ok %d (%zu images, centre %.9g %.9g %.9g)\n", i,
                    (int)okA[i], A[i].images_.size(), A[i].center_[0], A[i].center_[1], A[i].center_[2], (int)okB[i], B[i].images_.size(),
                    B[i].center_[0], B[i].center_[1], B[i].center_[2]);
        }
        good += okA[i];
    }
    // (3) a lone call
    const int reps = np < 40 ? np : 40;
    std::vector<double> lat;
    for (int i = 0; i < reps; i++) {
        mo3d::Patch3d c(seeds[i]);
        t0 = now();
        one.optimize(c);
        lat.push_back(now() - t0);
    }
    double lmin = 1e9, lsum = 0;
    for (double v : lat) { lmin = v < lmin ? v : lmin; lsum += v; }
    printf("{\"patches\": %d, \"threads\": %d, \"refined\": %d, \"mismatches\": %d, \"batch_s\": %.6f, \"batch_patches_per_s\": %.1f, "
           "\"calls_s\": %.6f, \"calls_patches_per_s\": %.1f, \"calls_over_batch\": %.4f, \"single_call_ms_min\": %.3f, "
           "\"single_call_ms_mean\": %.3f, \"loop_cpu_user_s\": %.4f, \"loop_cpu_sys_s\": %.4f}\n",
           np, T, good, mismatches, t_batch, np / t_batch, t_calls, np / t_calls, (np / t_calls) / (np / t_batch), lmin * 1e3,
           lsum / reps * 1e3, cpu_user, cpu_sys);
    return mismatches ? 1 : 0;
}
