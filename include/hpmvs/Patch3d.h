// mo3d::Patch3d -- the I/O record of PatchOptimizer::optimize, field for field as in the reference
// (include/hpmvs/Patch3d.h:33-86, src/hpmvs/Patch3d.cpp:28-74).
#ifndef HPMVS_PATCH3D_H_
#define HPMVS_PATCH3D_H_
#include <atomic>
#include <cstdint>
#include <memory>
#include <vector>
#include <hpmvs/Vec.h>
namespace mo3d {
class Scene;
class Patch3d {
private:
    static std::atomic_uint_least32_t runningIdx;
    size_t uid_;
    inline uint32_t generateUid() { return runningIdx.fetch_add(1); }
public:
    Patch3d();
    Patch3d(const Patch3d& other);
    virtual ~Patch3d();
    Patch3d& operator=(const Patch3d& cSource);  // copies all but the uid
    Eigen::Vector4f center_;
    Eigen::Vector4f normal_;
    std::vector<int> images_;  // attached images, [0] = reference image
    float scale_3dx_;
    float dscale_;
    float ncc_;
    int priorityReduction_;
    bool expanded_;
    float flatness_;
    Eigen::Vector3f color_;
    bool dirty_;
    inline float x() const { return center_[0]; }
    inline float y() const { return center_[1]; }
    inline float z() const { return center_[2]; }
    inline size_t uid() const { return uid_; }
};
typedef std::shared_ptr<Patch3d> Ppatch3d;
}  // namespace mo3d
#endif
