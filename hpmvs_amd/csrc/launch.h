// launch.h -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_types.h"

namespace hpmvs {

void launch_half_resize(const uint8_t* src, int w, int h, uint8_t* dst, hipStream_t st);
void launch_objective(const DevScene& sc, const DevOptions& o, const DevBatch& b, const double* xs, double* f_out,
                      int32_t* ngrabs_out, hipStream_t st);
// the same values from the one-lane-per-grab kernel (kernel_objective_lane.hip)
void launch_objective_lane(const DevScene& sc, const DevOptions& o, const DevBatch& b, const double* xs, double* f_out,
                           int32_t* ngrabs_out, hipStream_t st);
void launch_inccs(const DevScene& sc, const DevOptions& o, const DevBatch& b, int ref_idx, int robust, float* out,
                  hipStream_t st);
void launch_seed_init(const DevScene& sc, const DevOptions& o, int start_level, int n, const double* xyz,
                      const int32_t* meas_off, const int32_t* meas_img, const DevBatch& b, hipStream_t st);
void launch_drift_gate(int n, const double* xyz, const DevBatch& b, hipStream_t st);
// CellProcessor::extend / ::branch candidates (mode 0 / 1): parents -> out (n * N patches), and the gates after optimize
int expand_fanout(int mode);
const float* expand_direction_table(int mode);
void launch_expand_init(const DevScene& sc, int mode, int n, const DevBatch& parents, const float* cell_center,
                        const float* cell_width, const uint8_t* skip, const DevBatch& out, hipStream_t st);
void launch_expand_gate(int mode, int n, const DevBatch& parents, const float* cell_center, const float* cell_width,
                        const DevBatch& out, hipStream_t st);
// workspace: optimize_workspace_bytes(n_cus) bytes per in-flight launch; its first 1 KB (work-queue counter)
// must be zero when the kernel starts
size_t optimize_workspace_bytes(int n_cus);
void launch_optimize(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace,
                     int n_cus, hipStream_t st);
// open batch (DevBatch::svc_ctrl set, arrays host-mapped): `wgs` persistent workgroups, workspace of
// optimize_workspace_bytes_for(wgs) bytes with a zeroed first 1 KB
size_t optimize_workspace_bytes_for(int wgs);
size_t optimize_stage_bytes_for(int wgs);  // the slots' staging records (DevBatch::svc_stage)
void launch_optimize_service(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int wgs,
                             hipStream_t st);

// depth maps and the acceptance gates over them (kernel_depth.hip)
void launch_depth_fill(float* p, size_t n, hipStream_t st);
void launch_set_depths(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, hipStream_t st);
void launch_depth_gates(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, float margin, int abs_int,
                        int32_t* n_visible, int32_t* n_blocking, int32_t* n_free, hipStream_t st);

// Scene::setDepths(patch, subtract) in call order (keys -> sort -> per-cell replay) and Scene::getLevelSupport
void launch_level_support(const DevScene& sc, const DevBatch& b, int min_level, int32_t* support, hipStream_t st);
void launch_depth_ops_keys(const DevScene& sc, const DevDepthView* depths, const float* pool, const DevBatch& b, unsigned long long* keys,
                           unsigned int* counter, hipStream_t st);
int depth_ops_sort(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, unsigned int count, hipStream_t st);
void launch_depth_ops_apply(const DevScene& sc, const DevDepthView* depths, float* pool, const DevBatch& b, const uint8_t* subtract,
                            const unsigned long long* keys, unsigned int count, hipStream_t st);

// the cells a batch's gates read and setDepths would write (the scheduler's conflict test; layouts at the kernel)
void launch_depth_footprints(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, int32_t* wr, int32_t* fr, int32_t* at,
                             int32_t* vb, hipStream_t st);

// refined-patch records of the multi-GPU exchange (include/hpmvs_amd.h: hpmvs_record, 192 bytes)
void launch_pack_records(const DevBatch& b, void* records, hipStream_t st);
void launch_unpack_records(const void* records, int n, const DevBatch& b, hipStream_t st);
// ... and the tails of the lists longer than a record's 64 ids (scratch_counts: one int per 64 patches, total: one device int)
void launch_pack_record_tails(const DevBatch& b, void* tails, int cap, int32_t* scratch_counts, int32_t* total, hipStream_t st);
void launch_unpack_record_tails(const void* tails, int n_tails, int patch_offset, const DevBatch& b, hipStream_t st);
// diagnostics: the BOBYQA state machine on analytic objectives (kernel_selftest.hip); device pointers
void launch_selftest(int n, const int* kind, const double* params, const double* x0, const double* lb, const double* ub,
                     int maxeval, double* xfinal, double* minf, int* rc, int* nevals, int* rescues, double* trace,
                     int trace_cap, double* cold, hipStream_t st);
size_t selftest_cold_doubles(int n);  // doubles of the `cold` scratch (the optimiser states' cold arrays) for n problems

}  // namespace hpmvs
