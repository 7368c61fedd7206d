// mo3d::HpmvsOptions -- same fields and defaults as the reference (include/hpmvs/HpmvsOptions.h:29-58).
#ifndef HPMVS_HPMVSOPTIONS_H_
#define HPMVS_HPMVSOPTIONS_H_
#include <cmath>
#include <string>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
namespace mo3d {
struct HpmvsOptions {
    // image pyramid
    int MAXLEVEL = 5;
    int MINLEVEL = 0;
    int START_LEVEL = 4;
    float MAX_ANGLE = 60.0f * M_PI / 180.0f;
    float MIN_ANGLE = 10.0f * M_PI / 180.0f;
    // tree (scheduler side; not read by the refinement path)
    bool FILTER_SCENE_CENTER = false;
    int PATCH_INIT_MAXLEVEL = 9;
    int MAX_TREE_LEVEL = 20;
    int PATCH_FINAL_MINLEVEL = 8;
    // optimization options
    int NR_OPTIMIZATION_THREADS = 3;
    int MAX_IMAGES_PER_PATCH = 6;
    int MIN_IMAGES_PER_PATCH = 3;
    float NCC_ALPHA_1 = 0.4;
    float NCC_ALPHA_2 = 0.5;
    float DEPTH_TEST_FACTOR = 1.0f;
    std::string OUTFOLDER = "/tmp";
};
}  // namespace mo3d
#endif
