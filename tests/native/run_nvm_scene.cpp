// End-to-end through real files (BASELINE.json configs[0], "tiny nvm scene"): read an NVM_V3 model whose
// cameras point at binary PPM images, build the Scene the way hp_pmvs() does (reference src/main.cpp:98-116:
// readFile -> addCameras -> extractCoVisiblilty -> initPatches), write the refined seeds as an extended PLY,
// and round-trip the NVM through saveNVM.
//   run_nvm_scene <in.nvm> <out.ply> <out.nvm> <start_level> [binary=1]
#include <cstdio>
#include <cstdlib>
#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/NVMReader.h>
#include <hpmvs/PlyWriter.h>
#include <hpmvs/Scene.h>

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    std::vector<mo3d::NVM_Model> models;
    mo3d::NVMReader::readFile(argv[1], models, true);
    if (models.empty()) { fprintf(stderr, "no model\n"); return 3; }
    mo3d::NVMReader::saveNVM(argv[3], models);
    mo3d::HpmvsOptions options;
    options.START_LEVEL = atoi(argv[4]);
    mo3d::Scene scene;
    if (!scene.addCameras(models[0], options)) { fprintf(stderr, "addCameras failed\n"); return 4; }
    if (!scene.extractCoVisiblilty(models[0], options)) return 5;
    std::vector<mo3d::Ppatch3d> patches;
    if (!scene.initPatches(models[0], options, patches)) { fprintf(stderr, "initPatches failed\n"); return 6; }
    const bool binary = argc < 6 || atoi(argv[5]) != 0;
    if (!mo3d::writeExtPly(argv[2], patches, binary, true, true, true)) return 7;
    printf("cameras %zu points %zu refined %zu covis0 %zu\n", models[0].cameras.size(), models[0].points.size(),
           patches.size(), scene.covis_[0].size());
    return 0;
}
