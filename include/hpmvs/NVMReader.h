// NVM model records the scene is built from (reference include/hpmvs/NVMReader.h:32-55).  Only the
// plain data types are mirrored here; the NVM_V3 text reader is host I/O outside the refinement path.
#ifndef HPMVS_NVMREADER_H_
#define HPMVS_NVMREADER_H_
#include <string>
#include <vector>
#include <hpmvs/Vec.h>
namespace mo3d {
struct NVM_Measurement { int imgIndex; int featIndex; double xy[2]; };
struct NVM_Point { Eigen::Vector3d xyz; Eigen::Vector3d rgb; std::vector<NVM_Measurement> measurements; };
struct NVM_Camera {
    std::string filename;
    double f;             // focal length
    Eigen::Vector4d rq;   // rotation quaternion <wxyz>
    Eigen::Vector3d c;    // camera center
    double r;             // radial distortion
};
struct NVM_Model { std::vector<NVM_Camera> cameras; std::vector<NVM_Point> points; };

// VisualSFM NVM_V3 text files (reference include/hpmvs/NVMReader.h:57-66, src/hpmvs/NVMReader.cpp:31-182):
//   NVM_V3 / <nCameras> / per camera "name f qw qx qy qz cx cy cz r 0" / <nPoints> /
//   per point "x y z r g b nMeas {img feat u v}" / ... / a model with 0 cameras ends the file.
class NVMReader {
public:
    // fixPath: make relative image names relative to the folder of the .nvm file
    static void readFile(const char* path, std::vector<NVM_Model>& models, bool fixPath = false);
    static void saveNVM(const char* path, const std::vector<NVM_Model>& models);
};
}  // namespace mo3d
#endif
