// dev_types.h -- HBM-resident scene tables shared by the host C-ABI layer and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace hpmvs {

constexpr int kMaxLevels = 8;
constexpr int kMaxImages = 64;   // = wavefront width: lane v <-> attached image v (the batch kernel's list rows)
constexpr int kWideImages = 256; // HPMVS_MAX_IMAGES: what the wide kernel (kernel_wide.hip) takes over beyond 64
constexpr int kTex = 7;          // PatchTex = Patch2d<7,float> (reference Patch2d.hpp:88)
constexpr int kTexPix = 49;

// One view: what PatchOptimizer reads of mo3d::Camera + mo3d::Image
// (reference Camera.h:87-105, Image.h:63-64,89-115).  The pyramid levels of a view are ONE allocation: level l of
// interleaved u8 RGB, 3*(y*W+x)+c, starts at pix[0] + off[l] (256-byte aligned) and is followed by at least 16 B of
// padding, so the 8-byte tap loads never leave the buffer.  pix[l] = pix[0] + off[l] is kept for the host and for
// kernels that want one level; the evaluator loads pix[0] and the offsets together with the rest of the view's table
// and selects by level, which leaves it without a load that depends on the level (prep_from_view).
struct alignas(16) DevView {
    const uint8_t* pix[kMaxLevels];
    uint32_t off[kMaxLevels];
    int32_t w[kMaxLevels];
    int32_t h[kMaxLevels];
    float P0[12];      // level 0; level l = diag(2^-l, 2^-l, 1) * P0 (Camera.cpp:55-63)
    float center[4];
    float xaxis[4];
    float yaxis[4];
    float zaxis[4];
    float fsum;        // kMat_[0](0,0) + kMat_[0](1,1)
    int32_t n_levels;  // projection_.size()
    uint32_t slab_bytes;  // size of the view's pyramid allocation (host bookkeeping)
    int32_t pad[1];
};

// One view's depth maps (reference Scene.h:74-76, Scene.cpp:74-80): per pyramid level an Eigen::MatrixXf(rows, cols),
// column-major (element (y, x) at y + x * rows), rows / cols = level size / DEPTH_SUBSAMPLE.
struct DevDepthView {
    float* d[kMaxLevels];
    int32_t rows[kMaxLevels];
    int32_t cols[kMaxLevels];
};

struct DevScene {
    const DevView* views;
    const int32_t* covis_off;  // n_views + 1
    const int32_t* covis;
    int32_t n_views;
    // Camera::getLeveli (Camera.cpp:92-99) rounds log2(r) to FLOAT before std::round: level > k <=> (float)log2(r) >= k + 0.5.
    // lvl_thr[k] = the smallest double r for which the host's log2 says so (bisection at start-up, capi.hip): the
    // device counts r >= lvl_thr[k] and needs no log2 -- and agrees with the reference in the ~1e-8-wide band below
    // 2^(k+1/2) where the float rounding of the logarithm already reaches k + 0.5.
    double lvl_thr[kMaxLevels];
};

// HpmvsOptions (reference HpmvsOptions.h:29-58) + host-precomputed thresholds so device and host
// agree on them bit for bit.
struct DevOptions {
    int32_t MAXLEVEL, MINLEVEL;
    float MAX_ANGLE, MIN_ANGLE;
    int32_t MAX_IMAGES_PER_PATCH, MIN_IMAGES_PER_PATCH;
    float NCC_ALPHA_1, NCC_ALPHA_2;
    double cos_max_d;  // cos((double)MAX_ANGLE): PatchOptimizer.cpp:484 (unqualified cos)
    float cos_max_f;   // std::cos(MAX_ANGLE): PatchOptimizer.cpp:129,238
    float sort_thr;    // 1 - cos(10 deg): PatchOptimizer.cpp:184
    float angle_scale; // M_PI / 48.0f: PatchOptimizer.cpp:398
};

// Batch arrays on the device (see hpmvs_patch_batch in include/hpmvs_amd.h)
// record layouts of an open batch (dwords).  IN: 0-2 centre, 3 scale, 4-6 normal, 7 n_images, 8-71 image ids.
// OUT: 0 ok, 1 stage, 2 nlopt result, 3 evaluations, 4 grabs, 5 n_images, 8-11 centre, 12-15 normal, 16-18 colour,
// 19 ncc, 20-21 fmin (f64), 22-27 x (f64 x 3), 32-95 image ids.
constexpr int kSvcInDwords = 80;
constexpr int kSvcOutDwords = 96;

struct DevBatch {
    int32_t n, max_images;
    float* center; float* normal; float* scale;
    int32_t* n_images; int32_t* images;
    uint8_t* ok; float* color; float* ncc;
    double* fmin; double* x; int32_t* result; int32_t* nevals; int32_t* stage; int32_t* ngrabs;
    // open batch (single-patch callers, capi.hip: Service): n = room of the batch; the host appends patches while the
    // kernel runs.  svc_in / svc_out: host-mapped records of kSvcInDwords / kSvcOutDwords dwords per patch, svc_ctrl[0]
    // = patches published so far, svc_ctrl[1] != 0: no more will come; the kernel sets svc_done[i] once patch i's
    // output record is visible to the host.  svc_stage: device-memory staging records, one per resident slot.
    const unsigned long long* svc_in; uint32_t* svc_out; uint32_t* svc_stage;
    const int32_t* svc_ctrl; int32_t* svc_done; int32_t svc_lanes;
};

}  // namespace hpmvs
