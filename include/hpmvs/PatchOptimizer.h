// mo3d::PatchOptimizer -- the drop-in boundary.  Same class name, namespace, constructor and
// optimize() as the reference (include/hpmvs/PatchOptimizer.h:39-43); copyable and cheap to
// construct (the reference holds one per thread in a std::vector: src/hpmvs/Scene.cpp:94-96,
// src/main.cpp:123-125).  optimize() is a batch of one through the C ABI; optimizeBatch() is the
// additive entry for callers that can hand over many independent patches at once (the seed loop of
// Scene::initPatches, Scene.cpp:114-167).
#ifndef HPMVS_PATCHOPTIMIZER_H
#define HPMVS_PATCHOPTIMIZER_H
#include <cstddef>
#include <cstdint>
#include <unordered_set>
#include <vector>
#include <hpmvs/Patch3d.h>
namespace mo3d {
class HpmvsOptions;
class Scene;
class PatchOptimizer {
public:
    PatchOptimizer(const mo3d::HpmvsOptions& options, const mo3d::Scene* scene);
    // false = drop this patch (patch untouched), exactly the reference's convention; infrastructure
    // errors (HIP failure, no device) also return false after logging to stderr.
    bool optimize(mo3d::Patch3d& patch);
    // ok[i] = result of optimize(*patches[i]); returns the number of successes
    size_t optimizeBatch(mo3d::Patch3d* const* patches, size_t n, uint8_t* ok);
    // Frontier expansion: the candidate loops of CellProcessor::extend (mode EXTEND, 6 candidates per
    // parent, reference src/hpmvs/CellProcessor.cpp:84-142) and CellProcessor::branch (mode BRANCH, 4
    // candidates, :210-262) for many cells in one GPU call: candidates are constructed exactly as the
    // reference does (*newP = *p, new centre/scale, expanded_ = false, flatness_ = -1), optimized, and
    // the geometric gates applied.  cells[i] = (leaf centre, leaf width) of parents[i]; skip (optional,
    // n*N flags) marks candidates whose target leaf the caller already found occupied (:120-124).
    // candidates receives n*N patches (candidate k of parent i at i*N + k), accepted[i*N + k] != 0 where
    // the candidate passed; the depth tests and the octree insertion stay with the caller.  Returns N.
    enum ExpandMode { EXTEND = 0, BRANCH = 1 };
    struct CellRef { Eigen::Vector3f c; float width; };
    int expandBatch(ExpandMode mode, const mo3d::Patch3d* const* parents, const CellRef* cells, size_t n,
                    const uint8_t* skip, std::vector<mo3d::Patch3d>& candidates, std::vector<uint8_t>& accepted);
    // One priority level of CellProcessor::extend (reference src/hpmvs/CellProcessor.cpp:84-178; main.cpp:146-181 pops the leaves of
    // one priority and runs it on each, one after the other) as a BATCHED frontier with the reference's SEQUENTIAL result: ONE
    // expandBatch for the level, then conflict-free waves of depthGates / walk in the reference's order / setDepths (a candidate is
    // decided unless a map cell it reads or would write, or its leaf, still depends on an undecided earlier candidate -- then it
    // waits for the next wave; hpmvs_amd/frontier.py is the same walk in Python, INTEGRATION.md has the argument).  parents: the
    // leaves' patches in the scheduler's order; width: the leaves' width; `occupied`: the scheduler's occupancy as a set of leaf
    // keys, updated in place (the octree itself stays with the scheduler: `leafKey` maps a point to its leaf, default = the
    // uniform grid floor(p / width)); the scene's depth maps (Scene::resetDepths / setDepths) receive the accepted candidates.
    // stage: 0 accepted and inserted, 20 leaf already taken (no refinement), 1 refinement or the scale / drift gates failed,
    // 23 / 24 / 25 depthTests / viewBlockTest / pixelFreeTests, 26 addConditional found the refined patch's leaf taken.
    struct LevelResult {
        std::vector<mo3d::Patch3d> candidates;   // 6 per parent
        std::vector<int> stage;
        std::vector<int> counts;                 // 3 per candidate: the counts at decision time (-1: not reached)
        std::vector<size_t> accepted;            // candidate indices, the reference's order
        int waves = 0;
    };
    typedef uint64_t (*LeafKeyFn)(const Eigen::Vector3f& p, float width, void* user);
    bool extendLevel(const mo3d::Patch3d* const* parents, size_t n, float width, std::unordered_set<uint64_t>& occupied,
                     float margin, bool absInt, LevelResult& out, bool sequential = true, LeafKeyFn leafKey = nullptr, void* user = nullptr);
    // One priority level of CellProcessor::branch (reference src/hpmvs/CellProcessor.cpp:210-307) over the patches of the level's
    // leaves, in the scheduler's order: level-support gate (:221-224), the four diagonal children with Cell::contains before and
    // after optimize (:233-258) as ONE expandBatch, then the depth maps in the reference's order -- per split leaf its patch taken
    // back (:276-279), its children entered (:296) -- as ONE ordered Scene::setDepths call.  finalLevel[i]: nodeLevel(leaf i) >=
    // PATCH_FINAL_MINLEVEL (the scheduler's knowledge): such a leaf keeps its patch when no child survived (:265-266).  The tree
    // operations (split, the children's leaves, the queue) stay with the caller; the result says what to do with every leaf.
    struct BranchResult {
        std::vector<mo3d::Patch3d> candidates;   // 4 per leaf
        std::vector<uint8_t> child;              // 4 per leaf: goes into the new leaves
        std::vector<int> support;                // Scene::getLevelSupport of the leaf's patch
        std::vector<uint8_t> split;              // per leaf: split (its patch's depths are out of the maps, the children's in)
    };
    bool branchLevel(const mo3d::Patch3d* const* parents, const CellRef* cells, size_t n, const uint8_t* finalLevel, BranchResult& out);
    // diagnostics of the last optimize()/optimizeBatch() call that the reference computes and drops
    // (final mean robust INCC f*, PatchOptimizer.cpp:365,376): one entry per patch
    const std::vector<double>& lastObjective() const { return lastF_; }
    const std::vector<int>& lastEvaluations() const { return lastEvals_; }
    // The batch entries keep their structure-of-arrays copies in pinned host memory that the GPU uses in place, reused from
    // call to call through a per-thread cache.  Bytes the CALLING thread holds for reuse right now, and the cap beyond which
    // returned blocks go back to the system (default 512 MB, $HPMVS_PIN_CACHE_MB; setPinnedCacheCap changes it for the
    // calling thread and trims at once).
    static size_t pinnedCacheBytes();
    static size_t pinnedCacheCap();
    static void setPinnedCacheCap(size_t bytes);
private:
    const mo3d::HpmvsOptions* options_p;
    const mo3d::Scene* scene_p;
    std::vector<double> lastF_;
    std::vector<int> lastEvals_;
    int preferFullRows_ = 0;   // small calls that go straight to rows as wide as a list can get (a dense scene: optimizeBatch)
};
}  // namespace mo3d
#endif
