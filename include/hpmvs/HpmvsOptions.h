// mo3d::HpmvsOptions -- the option record the reference's callers pass to PatchOptimizer / Scene /
// CellProcessor (reference include/hpmvs/HpmvsOptions.h:29-58).  Same public member names and default
// values, so code written against the reference compiles unchanged; the members are grouped by who
// reads them here.
#ifndef HPMVS_HPMVSOPTIONS_H_
#define HPMVS_HPMVSOPTIONS_H_
#include <string>
namespace mo3d {
class HpmvsOptions {
public:
    // ---- read by the refinement path (PatchOptimizer, and through it the GPU kernels)
    int MAXLEVEL, MINLEVEL;                              // pyramid levels built / lowest level used
    float MAX_ANGLE, MIN_ANGLE;                          // radians
    int MIN_IMAGES_PER_PATCH, MAX_IMAGES_PER_PATCH;      // the latter is never read by the reference either
    float NCC_ALPHA_1, NCC_ALPHA_2;                      // filterImagesNCC thresholds before / after BOBYQA
    // ---- read by the seed loop (Scene::initPatches)
    int START_LEVEL;
    // ---- read only by the scheduler side of the reference (octree, depth tests, threading, output)
    bool FILTER_SCENE_CENTER;
    int PATCH_INIT_MAXLEVEL, MAX_TREE_LEVEL, PATCH_FINAL_MINLEVEL;
    int NR_OPTIMIZATION_THREADS;
    float DEPTH_TEST_FACTOR;
    std::string OUTFOLDER;

    HpmvsOptions()
        : MAXLEVEL(5), MINLEVEL(0),
          MAX_ANGLE(static_cast<float>(60.0f * 3.14159265358979323846 / 180.0f)),
          MIN_ANGLE(static_cast<float>(10.0f * 3.14159265358979323846 / 180.0f)),
          MIN_IMAGES_PER_PATCH(3), MAX_IMAGES_PER_PATCH(6), NCC_ALPHA_1(0.4f), NCC_ALPHA_2(0.5f),
          START_LEVEL(4),
          FILTER_SCENE_CENTER(false), PATCH_INIT_MAXLEVEL(9), MAX_TREE_LEVEL(20), PATCH_FINAL_MINLEVEL(8),
          NR_OPTIMIZATION_THREADS(3), DEPTH_TEST_FACTOR(1.0f), OUTFOLDER("/tmp") {}
};
}  // namespace mo3d
#endif
