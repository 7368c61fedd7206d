// mo3d::Camera -- per-level projections and axes (reference include/hpmvs/Camera.h:28-107).
// Camera::init is the reference's host-side set-up (src/hpmvs/Camera.cpp:34-81); the float32 tables
// it produces are what gets uploaded to HBM.
#ifndef HPMVS_CAMERA_H_
#define HPMVS_CAMERA_H_
#include <string>
#include <vector>
#include <hpmvs/NVMReader.h>
#include <hpmvs/Vec.h>
namespace mo3d {
class Camera {
public:
    struct Mat34 { float m[12]; float operator()(int r, int c) const { return m[r * 4 + c]; } };
    struct Mat33 { float m[9]; float operator()(int r, int c) const { return m[r * 3 + c]; } };
    Camera() {}
    virtual ~Camera() {}
    void init(const mo3d::NVM_Camera* cam, int width, int height, const int maxLevel = 1);
    Eigen::Vector3f project(const Eigen::Vector4f& coord, const int level) const;
    int getLevels() const { return (int)projection_.size(); }
    float getScale(const Eigen::Vector4f& coord, const int level) const;
    float getLevel(const Eigen::Vector4f& coord, const float scale) const;
    int getLeveli(const Eigen::Vector4f& coord, const float scale, const int maxLevel) const;
    std::string name_;
    std::vector<Mat34> projection_;  // per level
    std::vector<Mat33> kMat_;        // per level
    Eigen::Vector4f center_;
    Eigen::Vector4f oAxis_;
    Eigen::Vector3f xAxis_, yAxis_, zAxis_;
    float ipscale_;
};
}  // namespace mo3d
#endif
