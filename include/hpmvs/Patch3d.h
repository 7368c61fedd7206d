// mo3d::Patch3d -- the record PatchOptimizer::optimize reads and writes (reference
// include/hpmvs/Patch3d.h:33-86, src/hpmvs/Patch3d.cpp:28-74).  Same public member names and semantics
// (a copy gets a fresh uid, assignment copies everything but the uid), so the reference's callers compile
// unchanged; members are grouped by meaning here.
#ifndef HPMVS_PATCH3D_H_
#define HPMVS_PATCH3D_H_
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>
#include <hpmvs/Vec.h>
namespace mo3d {
class Patch3d {
public:
    // ---- geometry: what optimize() refines (homogeneous: center_.w = 1, normal_.w = 0)
    Eigen::Vector4f center_, normal_;
    float scale_3dx_;          // patch extent in scene units per texture sample; optimize() leaves it unchanged
    // ---- visibility: ids of the attached images, images_[0] is the reference image
    std::vector<int> images_;
    // ---- appearance, filled by optimize() on success
    Eigen::Vector3f color_;
    float ncc_;
    // ---- bookkeeping of the reference's scheduler (octree / CellProcessor); not read by optimize()
    int priorityReduction_;
    bool expanded_, dirty_;
    float flatness_, dscale_;

    Patch3d();
    Patch3d(const Patch3d& other);            // new uid, then *this = other
    Patch3d& operator=(const Patch3d& rhs);   // everything except the uid
    virtual ~Patch3d();

    float x() const { return center_[0]; }
    float y() const { return center_[1]; }
    float z() const { return center_[2]; }
    size_t uid() const { return uid_; }

private:
    size_t uid_;
    static std::atomic_uint_least32_t runningIdx;  // source of the uids
    static uint32_t generateUid() { return runningIdx.fetch_add(1); }
};
typedef std::shared_ptr<Patch3d> Ppatch3d;
}  // namespace mo3d
#endif
