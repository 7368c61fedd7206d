// mo3d::Scene -- the part of the reference's Scene the refinement path reads (reference
// include/hpmvs/Scene.h:69-71: cameras_, images_, covis_; cached by PatchOptimizer at
// src/hpmvs/PatchOptimizer.cpp:38-41).  The octree and the scheduler stay with the host application; the
// depth maps (reference Scene.h:74-76) live in HBM next to the pyramids, with batch forms of setDepths and of the
// three acceptance tests the expansion gates its candidates on.  The scene must be complete before the first PatchOptimizer is constructed and is
// immutable afterwards, exactly as in the reference; at that point it is uploaded to HBM once and
// shared by every optimizer instance.
#ifndef HPMVS_SCENE_H_
#define HPMVS_SCENE_H_
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <hpmvs/Camera.h>
#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/Image.h>
#include <hpmvs/NVMReader.h>
#include <hpmvs/Patch3d.h>
struct hpmvs_scene;
namespace mo3d {
class Scene {
public:
    Scene();
    virtual ~Scene();
    // Scene::addCameras (src/hpmvs/Scene.cpp:42-88) minus the depth maps: Image::load + Camera::init
    bool addCameras(const NVM_Model& model, const HpmvsOptions& options);
    // Scene::extractCoVisiblilty (src/hpmvs/Scene.cpp:241-298), including its positional-index quirk
    bool extractCoVisiblilty(const NVM_Model& model, const HpmvsOptions& options);
    // The seed loop of Scene::initPatches (src/hpmvs/Scene.cpp:112-178) as ONE batched GPU call: seed
    // construction, optimize(), drift gate.  Survivors are appended to `out` in point order; inserting
    // them into the octree / depth maps (Scene.cpp:183-199) stays with the host application.
    bool initPatches(const NVM_Model& model, const HpmvsOptions& options, std::vector<Ppatch3d>& out) const;
    // ---- depth maps (src/hpmvs/Scene.cpp:74-80) and the acceptance tests over them, batched on the device.
    // resetDepths: allocate / clear (MAX_DEPTH = 1000).  setDepths: Scene::setDepths(patch) (Scene.cpp:351-381) for
    // every patch.  depthGates: per patch the counts of Scene::depthTests, viewBlockTest and pixelFreeTests
    // (Scene.cpp:518-644) that CellProcessor compares with MIN_IMAGES_PER_PATCH (CellProcessor.cpp:134-142);
    // absInt selects C's abs(int) for the unqualified abs() at Scene.cpp:571 (default: the <cmath> overload).
    bool resetDepths() const;
    bool setDepths(const Patch3d* const* patches, size_t n) const;
    // Scene::setDepths(patch, subtract) for a list of calls IN ORDER (src/hpmvs/Scene.cpp:351-381): subtract[i] != 0 takes patch i's
    // depths back (a cell that still holds exactly its depth becomes MAX_DEPTH again, :373-374 -- CellProcessor::branch does that
    // for the patch of a leaf it splits, CellProcessor.cpp:276-279).  That does not commute with the minimum of an ordinary call,
    // so the list is applied cell by cell in call order on the device: the maps are those of the sequential loop.
    bool setDepths(const Patch3d* const* patches, size_t n, const uint8_t* subtract) const;
    // Scene::getLevelSupport(patch, minLevel) (src/hpmvs/Scene.cpp:334-343) for every patch
    bool levelSupport(const Patch3d* const* patches, size_t n, int minLevel, std::vector<int>& support) const;
    bool depthGates(const Patch3d* const* patches, size_t n, float margin, std::vector<int>& nVisible,
                    std::vector<int>& nBlocking, std::vector<int>& nFree, bool absInt = false) const;
    // The map cells those tests READ and the cells setDepths would WRITE, per patch, as packed keys -- what a scheduler needs to
    // run one priority level in conflict-free waves and still end with the reference's sequential result
    // (PatchOptimizer::extendLevel).  reads[i]: every cell depthTests / viewBlockTest reach through getFullDepth (3x3 level-0
    // pixels, every pyramid level) and pixelFreeTests' cell; writes[i]: the cell per attached image.  nLevels: pyramid levels.
    bool depthFootprints(const Patch3d* const* patches, size_t n, std::vector<std::vector<uint64_t> >& reads,
                         std::vector<std::vector<uint64_t> >& writes, int nLevels = 6) const;
    std::map<std::string, int> dict_;
    std::vector<Camera> cameras_;
    std::vector<Image> images_;
    std::vector<std::vector<int> > covis_;
    // HBM-resident copy, created on first use (thread-safe), device = HPMVS_DEVICE env or 0
    hpmvs_scene* deviceScene() const;
    int device() const { return device_; }
    void setDevice(int d) { device_ = d; }
    // Combines concurrent single-patch PatchOptimizer::optimize() calls of several host threads (the reference's
    // callers: one optimizer per OpenMP thread, Scene.cpp:94-96,166-167, CellProcessor.cpp:129,256) into batched
    // device launches; opaque, owned by the scene
    void* combiner() const;
private:
    mutable hpmvs_scene* dev_;
    mutable void* combiner_ = nullptr;
    mutable std::mutex mu_;
    int device_;
    int maxLevel_;
};
}  // namespace mo3d
#endif
