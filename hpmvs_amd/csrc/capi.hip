// capi.hip -- the C ABI of include/hpmvs_amd.h: scene residency in HBM + batch launches.
// Host-side only; all arithmetic of the hot path is in the kernels.  There is no CPU fallback:
// without a HIP device every compute entry point fails with HPMVS_ERR_NODEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <time.h>
#include <vector>

#include "../../include/hpmvs_amd.h"
#include "dev_types.h"
#include "launch.h"

using namespace hpmvs;

// Each optimize launch gets its own workspace (work-queue counter block + the per-slot image-list rows),
// handed out round-robin, so that launches issued concurrently from several host threads / streams (the
// reference runs one PatchOptimizer per OpenMP thread against one shared Scene: src/hpmvs/Scene.cpp:94-96,
// 166-167) never share one.  16 launches in flight per scene; each of them fills the GPU.
static constexpr int kQueueSlots = 16;
static constexpr size_t kQueueSlotBytes = 1024;
static constexpr int kServiceMaxPatches = 4;  // host batches up to this size go through the open batch (Service)  // the counter block at the head of a workspace (zeroed per launch)

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(HPMVS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

struct hpmvs_scene {
    int device = 0;
    int n_views = 0;
    int n_cus = 256;
    bool committed = false;
    std::vector<DevView> hviews;
    std::vector<char> view_set;
    std::vector<std::vector<int32_t>> covis;
    std::vector<void*> allocs;
    DevView* dviews = nullptr;
    int32_t* dcovis_off = nullptr;
    int32_t* dcovis = nullptr;
    // open batches behind single-patch host calls (see Service below); two, so that one can drain while the other is open
    struct Service;
    mutable Service* svc[2] = {nullptr, nullptr};
    mutable int svc_cur = 0;
    mutable std::mutex svc_mu;
    // depth maps (hpmvs_scene_depth_reset): one pool, per view a table of level pointers
    mutable float* depth_pool = nullptr;
    mutable size_t depth_floats = 0;
    mutable DevDepthView* ddepth = nullptr;
    mutable std::vector<DevDepthView> hdepth;
    mutable int32_t* dqueue[kQueueSlots] = {};  // launch workspaces of ws_bytes each: the first at commit, the others on first concurrent use
    mutable int n_queues = 0;
    size_t ws_bytes = 0;
    mutable unsigned next_slot = 0;
    mutable int32_t* last_queue = nullptr;
    mutable std::atomic<unsigned long long> last_staged_bytes{0}, last_mapped_bytes{0};  // of the last host-pointer optimize_batch call
    size_t bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t slot_done[kQueueSlots] = {};   // recorded behind the last launch that used workspace k
    mutable bool slot_used[kQueueSlots] = {};
    mutable hipStream_t slot_stream[kQueueSlots] = {};   // the stream workspace k's last launch went to
    mutable std::recursive_mutex mu;  // (recursive: Exclusive scopes nest with the workspace hand-out)
    mutable bool timed = false;
    // Small host-pointer batches (the unchanged callers' optimize() = batch of 1, and what a combiner makes of
    // concurrent callers) are staged through persistent blocks: pinned host + device memory and a stream per
    // entry, one H2D and one D2H copy per call, no hipMalloc / hipFree
    struct StageEntry { hipStream_t st = nullptr; char* dev = nullptr; char* host = nullptr; std::mutex mu; };
    static constexpr int kStageEntries = 8;
    mutable int32_t* last_ws = nullptr;   // the workspace of the last refinement launch (hpmvs_last_wide_patches)
    static constexpr int kStageCapPatches = 4096;
    mutable StageEntry stage[kStageEntries];
    mutable std::atomic<unsigned> stage_next{0};
};

static int service_quiesce(const hpmvs_scene* s);
static void service_close_all(const hpmvs_scene* s);
static void service_free(hpmvs_scene* s);

// Workspace hand-out (call with s->mu held).  Before a workspace is reused the launch that used it last must
// have finished: more than kQueueSlots launches in flight against one scene (e.g. 256 OpenMP threads each
// calling optimize()) simply queue up here instead of sharing a counter / image-list rows.
static int acquire_workspace(const hpmvs_scene* s, int32_t** q, int* slot, hipStream_t st) {
    // The workspace whose last launch went to THIS stream is taken again without any wait: the counter-block memset and the
    // kernel of the new launch are ordered behind the old one by the stream itself (a scheduler that enqueues round after
    // round on one stream, like bench.py, never needs a second workspace).  Otherwise one whose last launch has finished;
    // only when every allocated one is still in flight on OTHER streams does the scene grow another (~210 MB each), and
    // with all kQueueSlots in flight the caller waits for the oldest.
    int k = -1;
    for (int i = 0; i < s->n_queues && k < 0; i++)
        if (s->slot_used[i] && s->slot_stream[i] == st) k = i;
    for (int i = 0; i < s->n_queues && k < 0; i++) {
        const int c = (int)((s->next_slot + i) % (unsigned)s->n_queues);
        if (!s->slot_used[c]) k = c;
        else if (hipEventQuery(s->slot_done[c]) == hipSuccess) k = c;
        else (void)hipGetLastError();  // hipErrorNotReady is not an error here
    }
    if (k < 0 && s->n_queues < kQueueSlots) {
        int32_t* w = nullptr;
        HIPCHK(hipMalloc((void**)&w, s->ws_bytes));
        // cleared ON THE LAUNCH STREAM: hipMemset on the null stream returns before the fill has run and is not ordered with the
        // non-blocking streams small host batches are launched on -- the fill of a workspace that was grown under concurrent
        // single-patch calls ran INTO the first launch that used it and zeroed its image-list rows (round 6: one call in a few
        // hundred came back with another refinement; tests/test_gpu_concurrent_first_calls.py)
        if (hipMemsetAsync(w, 0, s->ws_bytes, st) != hipSuccess) { hipFree(w); return fail(HPMVS_ERR_HIP, "cannot clear a launch workspace"); }
        k = s->n_queues;
        s->dqueue[k] = w;
        s->n_queues++;
        const_cast<hpmvs_scene*>(s)->bytes += s->ws_bytes;
    }
    if (k < 0) {
        k = (int)(s->next_slot % kQueueSlots);
        HIPCHK(hipEventSynchronize(s->slot_done[k]));
    }
    // whatever path chose it: the new launch is ordered behind the workspace's previous one on the device (a no-op for the
    // same stream and for a finished launch; it also covers a stream handle that was destroyed and handed out again)
    if (s->slot_used[k]) HIPCHK(hipStreamWaitEvent(st, s->slot_done[k], 0));
    s->next_slot = (unsigned)k + 1;
    s->slot_stream[k] = st;
    *q = s->dqueue[k];
    *slot = k;
    s->last_queue = *q;
    return HPMVS_OK;
}
static int release_workspace(const hpmvs_scene* s, int slot, hipStream_t st) {
    HIPCHK(hipEventRecord(s->slot_done[slot], st));
    s->slot_used[slot] = true;
    return HPMVS_OK;
}

extern "C" {

const char* hpmvs_last_error(void) { return g_err.c_str(); }
#ifndef HPMVS_BUILD_ID
#define HPMVS_BUILD_ID "unstamped"
#endif
const char* hpmvs_build_id(void) { return HPMVS_BUILD_ID; }

int hpmvs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void hpmvs_default_options(hpmvs_options* o) {
    // reference include/hpmvs/HpmvsOptions.h:29-58
    o->MAXLEVEL = 5;
    o->MINLEVEL = 0;
    o->MAX_ANGLE = (float)(60.0f * M_PI / 180.0f);
    o->MIN_ANGLE = (float)(10.0f * M_PI / 180.0f);
    o->MAX_IMAGES_PER_PATCH = 6;
    o->MIN_IMAGES_PER_PATCH = 3;
    o->NCC_ALPHA_1 = 0.4f;
    o->NCC_ALPHA_2 = 0.5f;
}

static DevOptions make_dev_options(const hpmvs_options* o) {
    DevOptions d;
    d.MAXLEVEL = o->MAXLEVEL; d.MINLEVEL = o->MINLEVEL;
    d.MAX_ANGLE = o->MAX_ANGLE; d.MIN_ANGLE = o->MIN_ANGLE;
    d.MAX_IMAGES_PER_PATCH = o->MAX_IMAGES_PER_PATCH; d.MIN_IMAGES_PER_PATCH = o->MIN_IMAGES_PER_PATCH;
    d.NCC_ALPHA_1 = o->NCC_ALPHA_1; d.NCC_ALPHA_2 = o->NCC_ALPHA_2;
    d.cos_max_d = cos((double)o->MAX_ANGLE);
    d.cos_max_f = std::cos(o->MAX_ANGLE);
    d.sort_thr = (float)(1.0f - cos(10.0 * M_PI / 180.0));
    d.angle_scale = (float)(M_PI / 48.0f);
    return d;
}

// Camera::init, reference src/hpmvs/Camera.cpp:34-81 (float32 after the float64 quaternion)
int hpmvs_camera_from_nvm(double f, const double q[4], const double c[3], int width, int height, int max_level,
                          hpmvs_camera* out) {
    if (!q || !c || !out || width < 1 || height < 1 || max_level < 0 || max_level + 1 > HPMVS_MAX_LEVELS)
        return fail(HPMVS_ERR_ARG, "camera_from_nvm: bad argument");
    const float K[9] = {(float)f, 0.f, (float)(width / 2.0), 0.f, (float)f, (float)(height / 2.0), 0.f, 0.f, 1.f};
    // Eigen::Quaterniond::matrix() in double (Camera.cpp:43-50), no normalisation
    const double tx = 2.0 * q[1], ty = 2.0 * q[2], tz = 2.0 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    const float R[9] = {(float)(1.0 - (tyy + tzz)), (float)(txy - twz), (float)(txz + twy),
                        (float)(txy + twz), (float)(1.0 - (txx + tzz)), (float)(tyz - twx),
                        (float)(txz - twy), (float)(tyz + twx), (float)(1.0 - (txx + tyy))};
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
    float Rt[12];
    for (int r = 0; r < 3; r++) {
        Rt[r * 4 + 0] = R[r * 3 + 0]; Rt[r * 4 + 1] = R[r * 3 + 1]; Rt[r * 4 + 2] = R[r * 3 + 2];
        Rt[r * 4 + 3] = -((R[r * 3 + 0] * cf[0] + R[r * 3 + 1] * cf[1]) + R[r * 3 + 2] * cf[2]);
    }
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++)
            out->P0[r * 4 + col] =
                (K[r * 3 + 0] * Rt[0 * 4 + col] + K[r * 3 + 1] * Rt[1 * 4 + col]) + K[r * 3 + 2] * Rt[2 * 4 + col];
    out->center[0] = cf[0]; out->center[1] = cf[1]; out->center[2] = cf[2]; out->center[3] = 1.0f;
    const float* r2 = &out->P0[8];
    const float nrm = std::sqrt((r2[0] * r2[0] + r2[1] * r2[1]) + r2[2] * r2[2]);
    float z[3] = {r2[0] / nrm, r2[1] / nrm, r2[2] / nrm};
    float x[3] = {out->P0[0], out->P0[1], out->P0[2]};
    auto cross = [](const float* a, const float* b, float* o) {
        const float o0 = a[1] * b[2] - a[2] * b[1], o1 = a[2] * b[0] - a[0] * b[2], o2 = a[0] * b[1] - a[1] * b[0];
        o[0] = o0; o[1] = o1; o[2] = o2;
    };
    auto normalize = [](float* a) {
        const float n2 = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
        if (n2 > 0.f) { const float n = std::sqrt(n2); a[0] /= n; a[1] /= n; a[2] /= n; }
    };
    float y[3];
    cross(z, x, y); normalize(y);
    cross(y, z, x); normalize(x);
    for (int k = 0; k < 3; k++) { out->xaxis[k] = x[k]; out->yaxis[k] = y[k]; out->zaxis[k] = z[k]; }
    out->fsum = K[0] + K[4];
    out->n_levels = max_level + 1;
    return HPMVS_OK;
}

int hpmvs_scene_create(int n_views, int device, hpmvs_scene** out) {
    if (!out || n_views <= 0 || n_views > 65535) return fail(HPMVS_ERR_ARG, "scene_create: bad n_views/out");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(HPMVS_ERR_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(HPMVS_ERR_ARG, "scene_create: bad device index");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    hpmvs_scene* s = new hpmvs_scene();
    struct Guard { hpmvs_scene* p; ~Guard() { if (p) { if (p->ev0) hipEventDestroy(p->ev0); if (p->ev1) hipEventDestroy(p->ev1); delete p; } } } guard{s};
    s->device = device;
    s->n_views = n_views;
    s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    s->hviews.resize(n_views);
    memset(s->hviews.data(), 0, sizeof(DevView) * (size_t)n_views);
    s->view_set.assign(n_views, 0);
    s->covis.resize(n_views);
    HIPCHK(hipEventCreate(&s->ev0));
    HIPCHK(hipEventCreate(&s->ev1));
    guard.p = nullptr;
    *out = s;
    return HPMVS_OK;
}

int hpmvs_scene_set_view(hpmvs_scene* s, int view, int width, int height, const uint8_t* rgb_l0, int rgb_on_device,
                         const hpmvs_camera* cam) {
    if (!s || !cam || !rgb_l0) return fail(HPMVS_ERR_ARG, "scene_set_view: null argument");
    if (s->committed) return fail(HPMVS_ERR_STATE, "scene_set_view: scene already committed");
    if (view < 0 || view >= s->n_views || width < 2 || height < 2) return fail(HPMVS_ERR_ARG, "scene_set_view: bad view/size");
    if (cam->n_levels < 1 || cam->n_levels > HPMVS_MAX_LEVELS) return fail(HPMVS_ERR_ARG, "scene_set_view: bad n_levels");
    HIPCHK(hipSetDevice(s->device));
    DevView& V = s->hviews[view];
    // Image::init/load: max(1, MAXLEVEL) + 1 pyramid levels (reference Image.cpp:33-57); the camera has
    // MAXLEVEL + 1 projection levels (Camera.cpp:36).  Both are MAXLEVEL + 1 for MAXLEVEL >= 1.
    const int levels = cam->n_levels < 2 ? 2 : cam->n_levels;
    // one allocation per view: level l at a 256-byte aligned offset, 16 B of padding behind every level
    size_t off[HPMVS_MAX_LEVELS] = {}, total = 0;
    {
        int w = width, h = height;
        for (int l = 0; l < levels; l++) {
            if (w < 1 || h < 1) return fail(HPMVS_ERR_ARG, "scene_set_view: image too small for the pyramid depth");
            off[l] = total;
            total += ((size_t)w * h * 3 + 16 + 255) & ~(size_t)255;
            w /= 2; h /= 2;
        }
        if (total > 0xffffffffull) return fail(HPMVS_ERR_ARG, "scene_set_view: image too large (a view's pyramid must stay below 4 GB)");
    }
    uint8_t* slab = nullptr;
    HIPCHK(hipMalloc((void**)&slab, total));
    if (s->view_set[view]) {  // set again: release the previous pyramid of this view
        void* old = (void*)V.pix[0];
        for (size_t k = 0; k < s->allocs.size(); k++)
            if (s->allocs[k] == old) { s->allocs.erase(s->allocs.begin() + k); break; }
        if (old) { hipFree(old); s->bytes -= V.slab_bytes; }
        for (int l = 0; l < HPMVS_MAX_LEVELS; l++) V.pix[l] = nullptr;
        s->view_set[view] = 0;
    }
    s->allocs.push_back(slab);
    s->bytes += total;
    V.slab_bytes = (uint32_t)total;
    HIPCHK(hipMemset(slab, 0, total));
    int w = width, h = height;
    for (int l = 0; l < levels; l++) {
        uint8_t* d = slab + off[l];
        const size_t nb = (size_t)w * h * 3;
        if (l == 0) {
            HIPCHK(hipMemcpy(d, rgb_l0, nb, rgb_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        } else {
            launch_half_resize(V.pix[l - 1], V.w[l - 1], V.h[l - 1], d, nullptr);
            HIPCHK(hipGetLastError());
        }
        V.pix[l] = d; V.off[l] = (uint32_t)off[l]; V.w[l] = w; V.h[l] = h;
        w /= 2; h /= 2;
    }
    HIPCHK(hipDeviceSynchronize());
    memcpy(V.P0, cam->P0, sizeof(float) * 12);
    memcpy(V.center, cam->center, sizeof(float) * 4);
    for (int k = 0; k < 3; k++) { V.xaxis[k] = cam->xaxis[k]; V.yaxis[k] = cam->yaxis[k]; V.zaxis[k] = cam->zaxis[k]; }
    V.xaxis[3] = V.yaxis[3] = V.zaxis[3] = 0.0f;
    V.fsum = cam->fsum;
    V.n_levels = levels;
    s->view_set[view] = 1;
    return HPMVS_OK;
}

int hpmvs_scene_set_covis(hpmvs_scene* s, int view, const int32_t* ids, int n) {
    if (!s || (n > 0 && !ids)) return fail(HPMVS_ERR_ARG, "scene_set_covis: null argument");
    if (s->committed) return fail(HPMVS_ERR_STATE, "scene_set_covis: scene already committed");
    if (view < 0 || view >= s->n_views || n < 0) return fail(HPMVS_ERR_ARG, "scene_set_covis: bad view");
    for (int k = 0; k < n; k++)
        if (ids[k] < 0 || ids[k] >= s->n_views) return fail(HPMVS_ERR_ARG, "scene_set_covis: id out of range");
    s->covis[view].assign(ids, ids + n);
    return HPMVS_OK;
}

int hpmvs_scene_commit(hpmvs_scene* s) {
    if (!s) return fail(HPMVS_ERR_ARG, "scene_commit: null scene");
    if (s->committed) return fail(HPMVS_ERR_STATE, "scene_commit: already committed");
    for (int v = 0; v < s->n_views; v++)
        if (!s->view_set[v]) return fail(HPMVS_ERR_STATE, "scene_commit: view " + std::to_string(v) + " not set");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMalloc((void**)&s->dviews, sizeof(DevView) * (size_t)s->n_views));
    HIPCHK(hipMemcpy(s->dviews, s->hviews.data(), sizeof(DevView) * (size_t)s->n_views, hipMemcpyHostToDevice));
    std::vector<int32_t> off(s->n_views + 1, 0), flat;
    for (int v = 0; v < s->n_views; v++) {
        off[v + 1] = off[v] + (int32_t)s->covis[v].size();
        flat.insert(flat.end(), s->covis[v].begin(), s->covis[v].end());
    }
    if (flat.empty()) flat.push_back(0);
    HIPCHK(hipMalloc((void**)&s->dcovis_off, sizeof(int32_t) * off.size()));
    HIPCHK(hipMemcpy(s->dcovis_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&s->dcovis, sizeof(int32_t) * flat.size()));
    HIPCHK(hipMemcpy(s->dcovis, flat.data(), sizeof(int32_t) * flat.size(), hipMemcpyHostToDevice));
    for (int k = 0; k < kQueueSlots; k++) HIPCHK(hipEventCreateWithFlags(&s->slot_done[k], hipEventDisableTiming));
    s->ws_bytes = (optimize_workspace_bytes(s->n_cus) + 255) & ~(size_t)255;
    HIPCHK(hipMalloc((void**)&s->dqueue[0], s->ws_bytes));  // the others on first concurrent use (acquire_workspace)
    HIPCHK(hipMemset(s->dqueue[0], 0, s->ws_bytes));
    HIPCHK(hipStreamSynchronize(nullptr));   // (the fills of the upload have run before any launch on another stream can start)
    s->n_queues = 1;
    s->bytes += s->ws_bytes;
    s->bytes += sizeof(DevView) * (size_t)s->n_views + sizeof(int32_t) * (off.size() + flat.size());
    s->committed = true;
    return HPMVS_OK;
}

int hpmvs_scene_destroy(hpmvs_scene* s) {
    if (!s) return HPMVS_OK;
    hipSetDevice(s->device);
    service_free(s);  // closes and drains the open batches of single-patch calls
    for (void* p : s->allocs) hipFree(p);
    if (s->depth_pool) hipFree(s->depth_pool);
    if (s->ddepth) hipFree(s->ddepth);
    if (s->dviews) hipFree(s->dviews);
    if (s->dcovis_off) hipFree(s->dcovis_off);
    if (s->dcovis) hipFree(s->dcovis);
    for (int k = 0; k < s->n_queues; k++) if (s->dqueue[k]) hipFree(s->dqueue[k]);
    for (auto& e : s->stage) {
        if (e.st) hipStreamDestroy(e.st);
        if (e.dev) hipFree(e.dev);
        if (e.host) hipHostFree(e.host);
    }
    for (int k = 0; k < kQueueSlots; k++) if (s->slot_done[k]) hipEventDestroy(s->slot_done[k]);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    delete s;
    return HPMVS_OK;
}

int hpmvs_scene_get_level(const hpmvs_scene* s, int view, int level, uint8_t* host_out, size_t cap, int* w, int* h) {
    if (!s || !w || !h) return fail(HPMVS_ERR_ARG, "scene_get_level: null argument");
    if (view < 0 || view >= s->n_views || !s->view_set[view]) return fail(HPMVS_ERR_ARG, "scene_get_level: bad view");
    const DevView& V = s->hviews[view];
    if (level < 0 || level >= V.n_levels) return fail(HPMVS_ERR_ARG, "scene_get_level: bad level");
    *w = V.w[level]; *h = V.h[level];
    const size_t nb = (size_t)V.w[level] * V.h[level] * 3;
    if (host_out) {
        if (cap < nb) return fail(HPMVS_ERR_ARG, "scene_get_level: buffer too small");
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipMemcpy(host_out, V.pix[level], nb, hipMemcpyDeviceToHost));
    }
    return HPMVS_OK;
}

size_t hpmvs_scene_bytes(const hpmvs_scene* s) { return s ? s->bytes : 0; }

int hpmvs_build_pyramid(int device, const uint8_t* src, int w, int h, uint8_t* dst, int on_device) {
    if (!src || !dst || w < 2 || h < 2) return fail(HPMVS_ERR_ARG, "build_pyramid: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HPMVS_ERR_NODEVICE, "no HIP device visible");
    HIPCHK(hipSetDevice(device));
    const size_t nb = (size_t)w * h * 3, nb2 = (size_t)(w / 2) * (h / 2) * 3;
    if (on_device) {
        launch_half_resize(src, w, h, dst, nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        return HPMVS_OK;
    }
    uint8_t *ds = nullptr, *dd = nullptr;
    struct Free { uint8_t*& a; uint8_t*& b; ~Free() { if (a) hipFree(a); if (b) hipFree(b); } } free_on_exit{ds, dd};
    HIPCHK(hipMalloc((void**)&ds, nb));
    HIPCHK(hipMalloc((void**)&dd, nb2 + 16));
    HIPCHK(hipMemcpy(ds, src, nb, hipMemcpyHostToDevice));
    launch_half_resize(ds, w, h, dd, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(dst, dd, nb2, hipMemcpyDeviceToHost));
    return HPMVS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// batch staging: host arrays <-> device arrays

namespace {

// hipMalloc / hipFree (and friends) wait for the device to be idle -- which it never is while single-patch callers keep an
// open batch running (a second one starts while the first drains).  Whatever allocates or frees while a scene is in use
// does it in an Exclusive scope: the scene lock (nobody can start an open batch meanwhile, service_open needs it) with
// the open batches ended and drained.  A host-staged call is in such a scope twice, briefly: for its allocations and (in
// ~Staged) for its frees; its copies, its launch and its wait run outside, so that concurrent big host batches against
// one scene overlap.
struct Exclusive {
    const hpmvs_scene* s;
    explicit Exclusive(const hpmvs_scene* s_) : s(s_) { if (s) { s->mu.lock(); (void)service_quiesce(s); } }
    ~Exclusive() { if (s) s->mu.unlock(); }
    Exclusive(const Exclusive&) = delete;
    Exclusive& operator=(const Exclusive&) = delete;
};

struct Staged {
    DevBatch d;
    std::vector<void*> tmp;
    struct Fill { void* dst; const void* src; size_t bytes; };  // src == nullptr: zero-fill
    std::vector<Fill> fills;  // what stage_in allocated still has to be filled: stage_flush, outside the Exclusive scope
    bool staged = false;
    size_t staged_bytes = 0, mapped_bytes = 0;  // arrays copied through device buffers / used in place (pinned host memory)
    const hpmvs_scene* scene = nullptr;  // set by calls against a scene: the buffers are freed in an Exclusive scope
    ~Staged() {
        if (tmp.empty()) return;
        Exclusive ex(scene);
        for (void* p : tmp) hipFree(p);
    }
};

// Device address of [host, host + bytes) when the whole range lies in pinned host memory that is mapped into the GPU's
// address space (hipHostMalloc / hipHostRegister; torch's pin_memory()), else nullptr.  Such an array of a host-pointer
// call is used IN PLACE: the kernels read a patch's inputs once and write its outputs once, so they stream over PCIe while
// the kernel runs instead of costing a hipMalloc, two copies and a hipFree around it (C2: 222.3 -> 213.1 ms host to host
// against 212.1 ms of kernel).  HPMVS_ZERO_COPY=0 switches the detection off.
static void* mapped_host(const void* host, size_t bytes) {
    static const bool enabled = [] { const char* e = getenv("HPMVS_ZERO_COPY"); return !(e && e[0] == '0'); }();
    if (!enabled || !host || bytes == 0) return nullptr;
    void *d0 = nullptr, *d1 = nullptr;
    if (hipHostGetDevicePointer(&d0, const_cast<void*>(host), 0) != hipSuccess ||
        hipHostGetDevicePointer(&d1, const_cast<char*>((const char*)host + bytes - 1), 0) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error here
        return nullptr;
    }
    if ((const char*)d1 - (const char*)d0 != (ptrdiff_t)(bytes - 1)) return nullptr;
    // ... and both ends must belong to ONE allocation: two adjacent registrations could map their first and last byte the
    // right distance apart by accident.  Where the runtime cannot name the allocation the end-point test stands.
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d0) == hipSuccess && base && size) {
        if ((const char*)d0 < (const char*)base || (const char*)d0 + bytes > (const char*)base + size) return nullptr;
    } else {
        (void)hipGetLastError();
    }
    return d0;
}

// copy: the array is an input (its contents travel to the device).  need_zero: an output the kernel may leave untouched for
// some patches (e.g. `color` of a failed one), which a staged call returns as 0.  allow_mapped = false: `host` only stands
// for "non-null", the range behind it is not the caller's.
template <typename T>
int stage_in(Staged& st, T*& dst, const T* host, size_t count, bool copy, bool need_zero = true, bool allow_mapped = true) {
    dst = nullptr;
    if (!host || count == 0) return HPMVS_OK;
    if (allow_mapped) {
        if (void* m = mapped_host(host, sizeof(T) * count)) {
            dst = (T*)m;
            if (!copy && need_zero) st.fills.push_back({m, nullptr, sizeof(T) * count});
            st.mapped_bytes += sizeof(T) * count;
            return HPMVS_OK;
        }
    }
    st.staged_bytes += sizeof(T) * count;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(T) * count));
    st.tmp.push_back(p);
    st.fills.push_back({p, copy ? (const void*)host : nullptr, sizeof(T) * count});
    dst = (T*)p;
    return HPMVS_OK;
}

// The copies / zero-fills of everything stage_in allocated.  Separate from the allocations so that only those (and the
// frees) run in the Exclusive scope: two threads staging big host batches against one scene overlap their copies and
// their launches (16 workspaces), they only take turns at hipMalloc / hipFree.
// With a launch stream the fills are issued ON it (ordered with the launch, and not waiting behind an open-batch kernel the
// way null-stream work can: ADVICE r4); a pageable source is consumed before hipMemcpyAsync returns.
int stage_flush(Staged& st, hipStream_t on = nullptr) {
    bool null_fill = false;
    for (const Staged::Fill& f : st.fills) {
        if (on) {
            if (f.src) HIPCHK(hipMemcpyAsync(f.dst, f.src, f.bytes, hipMemcpyHostToDevice, on));
            else HIPCHK(hipMemsetAsync(f.dst, 0, f.bytes, on));
        } else if (f.src) { HIPCHK(hipMemcpy(f.dst, f.src, f.bytes, hipMemcpyHostToDevice)); null_fill = true; }
        else { HIPCHK(hipMemset(f.dst, 0, f.bytes)); null_fill = true; }
    }
    // a null-stream fill may return before it has run (hipMemset always, a copy from pageable memory once it is staged); the launch
    // that follows may go to a non-blocking stream of the caller's
    if (null_fill) HIPCHK(hipStreamSynchronize(nullptr));
    st.fills.clear();
    return HPMVS_OK;
}

template <typename T>
int stage_out(T* host, const T* dev, size_t count) {
    if (!host || !dev || count == 0) return HPMVS_OK;
    if ((const void*)dev == mapped_host(host, sizeof(T) * count)) return HPMVS_OK;  // used in place (stage_in)
    HIPCHK(hipMemcpy(host, dev, sizeof(T) * count, hipMemcpyDeviceToHost));
    return HPMVS_OK;
}

int check_batch_shape(const hpmvs_scene* s, const hpmvs_options* o, const hpmvs_patch_batch* b) {
    if (!s || !o || !b) return fail(HPMVS_ERR_ARG, "null scene/options/batch");
    if (!s->committed) return fail(HPMVS_ERR_STATE, "scene not committed");
    if (b->n < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "bad n/max_images");
    if (o->MAXLEVEL < 1 || o->MAXLEVEL >= HPMVS_MAX_LEVELS) return fail(HPMVS_ERR_ARG, "MAXLEVEL out of range");
    // a negative level would index the per-level tables with a negative subscript (addImages); an image-count floor
    // below 1 would let an empty list reach setINCCs (the reference's addImages fails on an empty list regardless)
    if (o->MINLEVEL < 0 || o->MINLEVEL > o->MAXLEVEL) return fail(HPMVS_ERR_ARG, "MINLEVEL out of range");
    if (o->MIN_IMAGES_PER_PATCH < 1) return fail(HPMVS_ERR_ARG, "MIN_IMAGES_PER_PATCH must be >= 1");
    // sampleTexture reads pyramid levels up to MAXLEVEL - 1 (PatchOptimizer.cpp:487); the device forms the level's address
    // arithmetically inside the view's slab (prep_from_view), so a view with fewer levels must be refused here
    for (int v = 0; v < s->n_views; v++)
        if (s->hviews[v].n_levels < o->MAXLEVEL) return fail(HPMVS_ERR_ARG, "options.MAXLEVEL exceeds the pyramid levels of a view");
    return HPMVS_OK;
}

int check_batch(const hpmvs_scene* s, const hpmvs_options* o, const hpmvs_patch_batch* b) {
    const int rc = check_batch_shape(s, o, b);
    if (rc) return rc;
    if (b->n > 0 && (!b->center || !b->normal || !b->scale || !b->n_images || !b->images))
        return fail(HPMVS_ERR_ARG, "batch input arrays missing");
    return HPMVS_OK;
}

// all_written: the kernel behind this call writes ok / fmin / x / result / nevals / stage / ngrabs of EVERY patch (the
// refinement kernel does: fetch and finish stages), so arrays used in place need no zero fill first
int stage_batch(Staged& st, const hpmvs_patch_batch* b, bool on_device, bool outputs, bool copy_inputs = true, bool all_written = false) {
    DevBatch& d = st.d;
    memset(&d, 0, sizeof(d));
    d.n = b->n; d.max_images = b->max_images;
    const size_t n = (size_t)b->n;
    if (on_device) {
        d.center = b->center; d.normal = b->normal; d.scale = b->scale; d.n_images = b->n_images; d.images = b->images;
        d.ok = b->ok; d.color = b->color; d.ncc = b->ncc; d.fmin = b->fmin; d.x = b->x; d.result = b->result;
        d.nevals = b->nevals; d.stage = b->stage; d.ngrabs = b->ngrabs;
        return HPMVS_OK;
    }
    st.staged = true;
    int rc;
    if ((rc = stage_in(st, d.center, b->center, 4 * n, copy_inputs))) return rc;
    if ((rc = stage_in(st, d.normal, b->normal, 4 * n, copy_inputs))) return rc;
    if ((rc = stage_in(st, d.scale, b->scale, n, copy_inputs))) return rc;
    if ((rc = stage_in(st, d.n_images, b->n_images, n, copy_inputs))) return rc;
    if ((rc = stage_in(st, d.images, b->images, n * b->max_images, copy_inputs))) return rc;
    if (outputs) {
        if ((rc = stage_in(st, d.ok, b->ok, n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.color, b->color, 3 * n, false))) return rc;
        if ((rc = stage_in(st, d.ncc, b->ncc, n, false))) return rc;
        if ((rc = stage_in(st, d.fmin, b->fmin, n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.x, b->x, 3 * n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.result, b->result, n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.nevals, b->nevals, n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.stage, b->stage, n, false, /*need_zero=*/!all_written))) return rc;
        if ((rc = stage_in(st, d.ngrabs, b->ngrabs, n, false, /*need_zero=*/!all_written))) return rc;
    }
    return HPMVS_OK;
}

int copy_outputs(hpmvs_patch_batch* b, const DevBatch& D) {
    const size_t n = (size_t)b->n;
    int rc;
    if ((rc = stage_out(b->n_images, D.n_images, n))) return rc;
    if ((rc = stage_out(b->images, D.images, n * b->max_images))) return rc;
    if ((rc = stage_out(b->ok, D.ok, n))) return rc;
    if ((rc = stage_out(b->color, D.color, 3 * n))) return rc;
    if ((rc = stage_out(b->ncc, D.ncc, n))) return rc;
    if ((rc = stage_out(b->fmin, D.fmin, n))) return rc;
    if ((rc = stage_out(b->x, D.x, 3 * n))) return rc;
    if ((rc = stage_out(b->result, D.result, n))) return rc;
    if ((rc = stage_out(b->nevals, D.nevals, n))) return rc;
    if ((rc = stage_out(b->stage, D.stage, n))) return rc;
    if ((rc = stage_out(b->ngrabs, D.ngrabs, n))) return rc;
    return HPMVS_OK;
}

DevScene dev_scene(const hpmvs_scene* s) {
    DevScene sc;
    sc.views = s->dviews; sc.covis_off = s->dcovis_off; sc.covis = s->dcovis; sc.n_views = s->n_views;
    // level thresholds (dev_types.h): smallest double r with (float)log2(r) >= k + 0.5, by bisection on the bit pattern
    static const std::vector<double> thr = [] {
        std::vector<double> t(kMaxLevels);
        for (int k = 0; k < kMaxLevels; k++) {
            const float target = (float)k + 0.5f;
            double lo = std::ldexp(1.0, k), hi = std::ldexp(1.0, k + 1);  // (float)log2(lo) = k < target <= k + 1 = (float)log2(hi)
            unsigned long long a, b;
            memcpy(&a, &lo, 8); memcpy(&b, &hi, 8);
            while (b - a > 1) {
                const unsigned long long m = a + (b - a) / 2;
                double r; memcpy(&r, &m, 8);
                if ((float)std::log2(r) >= target) b = m; else a = m;
            }
            memcpy(&t[k], &b, 8);
        }
        return t;
    }();
    for (int k = 0; k < kMaxLevels; k++) sc.lvl_thr[k] = thr[k];
    return sc;
}

}  // namespace

// One refinement launch on `st` (call with s->mu held): picks the kernel, hands out / zeroes its workspace and
// brackets it with the scene's event pair.
static int enqueue_refinement(const hpmvs_scene* s, const DevOptions& d, const DevBatch& batch, hipStream_t st) {
    int rc;
    // A batch launch needs whole CUs (one workgroup takes a CU's LDS), an open batch holds a quarter of every CU for as
    // long as single-patch callers keep it open -- and its long-running kernel may sit in the very hardware queue this
    // launch's packets go to.  So the open batches end first: closed (their published patches are still finished, a few
    // milliseconds) and drained.  Callers that come later start a new one, which queues behind this launch (service_open).
    if ((rc = service_quiesce(s))) return rc;
    {
        int32_t* q;
        int slot;
        if ((rc = acquire_workspace(s, &q, &slot, st))) return rc;
        // whatever happens below, the slot's event is recorded behind the work that was enqueued, so that the next
        // user of this workspace waits for it
        struct Release { const hpmvs_scene* s; int slot; hipStream_t st; ~Release() { hipEventRecord(s->slot_done[slot], st); s->slot_used[slot] = true; } } rel{s, slot, st};
        HIPCHK(hipMemsetAsync(q, 0, kQueueSlotBytes, st));
        HIPCHK(hipEventRecord(s->ev0, st));
        launch_optimize(dev_scene(s), d, batch, q, s->n_cus, st);
        s->last_ws = q;
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->ev1, st));
    }
    s->timed = true;
    return HPMVS_OK;
}
// ---- multi-GPU exchange ------------------------------------------------------------------------------------------
static_assert(sizeof(hpmvs_record) == 192, "hpmvs_record is 192 bytes");
static DevBatch as_dev_batch(const hpmvs_patch_batch* b) {
    DevBatch d;
    memset(&d, 0, sizeof(d));
    d.n = b->n; d.max_images = b->max_images;
    d.center = b->center; d.normal = b->normal; d.scale = b->scale; d.n_images = b->n_images; d.images = b->images;
    d.ok = b->ok; d.color = b->color; d.ncc = b->ncc; d.fmin = b->fmin; d.x = b->x; d.result = b->result;
    d.nevals = b->nevals; d.stage = b->stage; d.ngrabs = b->ngrabs;
    return d;
}
extern "C" int hpmvs_pack_records(const hpmvs_patch_batch* b, hpmvs_record* records, void* stream) {
    if (!b || b->n < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "pack_records: bad batch");
    if (b->n > 0 && (!records || !b->center || !b->normal || !b->scale || !b->n_images || !b->images))
        return fail(HPMVS_ERR_ARG, "pack_records: missing array");
    launch_pack_records(as_dev_batch(b), records, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return HPMVS_OK;
}
extern "C" int hpmvs_unpack_records(const hpmvs_record* records, int n, hpmvs_patch_batch* b, void* stream) {
    if (!b || n < 0 || b->n < n || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "unpack_records: bad batch");
    if (n > 0 && (!records || !b->center || !b->normal || !b->scale || !b->n_images || !b->images))
        return fail(HPMVS_ERR_ARG, "unpack_records: missing array");
    launch_unpack_records(records, n, as_dev_batch(b), (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return HPMVS_OK;
}
static_assert(sizeof(hpmvs_record_tail) == 392, "hpmvs_record_tail is 392 bytes");
extern "C" int hpmvs_pack_record_tails(const hpmvs_patch_batch* b, hpmvs_record_tail* tails, int32_t cap, int32_t* n_tails, void* stream) {
    if (!b || !n_tails || b->n < 0 || cap < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "pack_record_tails: bad batch");
    *n_tails = 0;
    if (b->n == 0 || b->max_images <= HPMVS_RECORD_IMAGES) return HPMVS_OK;   // (rows of up to 64 ids have no tails)
    if (!b->n_images || !b->images || (cap > 0 && !tails)) return fail(HPMVS_ERR_ARG, "pack_record_tails: missing array");
    hipStream_t st = (hipStream_t)stream;
    const size_t nblocks = ((size_t)b->n + 63) / 64;
    int32_t* scratch = nullptr;
    HIPCHK(hipMalloc((void**)&scratch, sizeof(int32_t) * (nblocks + 1)));
    launch_pack_record_tails(as_dev_batch(b), tails, cap, scratch, scratch + nblocks, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(n_tails, scratch + nblocks, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(scratch);
    if (e != hipSuccess) return fail(HPMVS_ERR_HIP, std::string("pack_record_tails: ") + hipGetErrorString(e));
    if (*n_tails > cap) return fail(HPMVS_ERR_ARG, "pack_record_tails: " + std::to_string(*n_tails) + " tails do not fit the " + std::to_string(cap) + " handed over");
    return HPMVS_OK;
}
extern "C" int hpmvs_unpack_record_tails(const hpmvs_record_tail* tails, int32_t n_tails, int32_t patch_offset, hpmvs_patch_batch* b, void* stream) {
    if (!b || n_tails < 0 || b->n < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "unpack_record_tails: bad batch");
    if (n_tails == 0) return HPMVS_OK;
    if (!tails || !b->images) return fail(HPMVS_ERR_ARG, "unpack_record_tails: missing array");
    launch_unpack_record_tails(tails, n_tails, patch_offset, as_dev_batch(b), (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return HPMVS_OK;
}
// ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream); ncclUint8 = 1, ncclSuccess = 0 (rccl.h)
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_allgather_fn find_allgather() {
    static nccl_allgather_fn fn = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        // HPMVS_RCCL_LIBRARY names the library to take ncclAllGather from (an application that links its own build of
        // RCCL under another name); otherwise the RCCL already in the process (the communicator's own), then the
        // system's librccl
        void* sym = nullptr;
        if (const char* named = getenv("HPMVS_RCCL_LIBRARY")) {
            if (void* h = dlopen(named, RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(h, "ncclAllGather");
        } else {
            sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
            if (!sym) {
                void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) sym = dlsym(h, "ncclAllGather");
            }
        }
        fn = (nccl_allgather_fn)sym;
    });
    return fn;
}
extern "C" int hpmvs_allgather_records(void* nccl_comm, const hpmvs_record* send, size_t count, hpmvs_record* recv, void* stream) {
    if (!nccl_comm || (count > 0 && (!send || !recv))) return fail(HPMVS_ERR_ARG, "allgather_records: null argument");
    nccl_allgather_fn fn = find_allgather();
    if (!fn) return fail(HPMVS_ERR_STATE, "allgather_records: RCCL (ncclAllGather) not found in the process, as librccl.so.1 nor as $HPMVS_RCCL_LIBRARY");
    if (count > 0) {
        // both buffers must be device memory of ONE device (the communicator's): a host pointer or a buffer of another
        // GPU would fault inside the collective, on every rank
        hipPointerAttribute_t as{}, ar{};
        if (hipPointerGetAttributes(&as, send) != hipSuccess || hipPointerGetAttributes(&ar, recv) != hipSuccess ||
            as.type != hipMemoryTypeDevice || ar.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            return fail(HPMVS_ERR_ARG, "allgather_records: send / recv must be device memory");
        }
        if (as.device != ar.device)
            return fail(HPMVS_ERR_ARG, "allgather_records: send is on device " + std::to_string(as.device) + ", recv on device " + std::to_string(ar.device));
    }
    const int r = fn(send, recv, count * sizeof(hpmvs_record), /*ncclUint8*/ 1, nccl_comm, (hipStream_t)stream);
    if (r != 0) return fail(HPMVS_ERR_HIP, "ncclAllGather failed with ncclResult_t " + std::to_string(r));
    return HPMVS_OK;
}

extern "C" int hpmvs_selftest_bobyqa(int device, int n, const int32_t* kind, const double* params, const double* x0,
                                     const double* lb, const double* ub, int maxeval, double* xfinal, double* minf,
                                     int32_t* result, int32_t* nevals, int32_t* rescue_calls, double* trace, int trace_cap) {
    if (n < 0 || (n > 0 && (!kind || !params || !x0 || !lb || !ub || !xfinal || !minf || !result || !nevals || !rescue_calls)))
        return fail(HPMVS_ERR_ARG, "selftest_bobyqa: missing array");
    if (trace && trace_cap < 1) return fail(HPMVS_ERR_ARG, "selftest_bobyqa: trace_cap");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HPMVS_ERR_NODEVICE, "no HIP device visible");
    if (n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(device));
    Staged sg;  // frees its device buffers on every exit path
    int* dkind = nullptr; double *dpar = nullptr, *dx0 = nullptr, *dlb = nullptr, *dub = nullptr, *dxf = nullptr, *dmin = nullptr, *dtr = nullptr;
    int *drc = nullptr, *dne = nullptr, *dres = nullptr;
    int rc;
    const size_t N = (size_t)n;
    if ((rc = stage_in(sg, dkind, (const int*)kind, N, true))) return rc;
    if ((rc = stage_in(sg, dpar, params, 8 * N, true))) return rc;
    if ((rc = stage_in(sg, dx0, x0, 3 * N, true))) return rc;
    if ((rc = stage_in(sg, dlb, lb, 3 * N, true))) return rc;
    if ((rc = stage_in(sg, dub, ub, 3 * N, true))) return rc;
    if ((rc = stage_in(sg, dxf, (const double*)xfinal, 3 * N, false))) return rc;
    if ((rc = stage_in(sg, dmin, (const double*)minf, N, false))) return rc;
    if ((rc = stage_in(sg, drc, (const int*)result, N, false))) return rc;
    if ((rc = stage_in(sg, dne, (const int*)nevals, N, false))) return rc;
    if ((rc = stage_in(sg, dres, (const int*)rescue_calls, N, false))) return rc;
    if (trace && (rc = stage_in(sg, dtr, (const double*)trace, 4 * N * (size_t)trace_cap, false))) return rc;
    double* dcold = nullptr;
    if ((rc = stage_in(sg, dcold, (const double*)xfinal /* (any non-null host pointer: zero-filled) */, selftest_cold_doubles(n), false, true, /*allow_mapped=*/false))) return rc;
    if ((rc = stage_flush(sg))) return rc;
    launch_selftest(n, dkind, dpar, dx0, dlb, dub, maxeval, dxf, dmin, drc, dne, dres, dtr, trace_cap, dcold, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if ((rc = stage_out(xfinal, dxf, 3 * N))) return rc;
    if ((rc = stage_out(minf, dmin, N))) return rc;
    if ((rc = stage_out((int*)result, drc, N))) return rc;
    if ((rc = stage_out((int*)nevals, dne, N))) return rc;
    if ((rc = stage_out((int*)rescue_calls, dres, N))) return rc;
    if (trace && (rc = stage_out(trace, dtr, 4 * N * (size_t)trace_cap))) return rc;
    return HPMVS_OK;
}



// ---- pooled staging of a small host batch -------------------------------------------------------------------------
namespace {
struct PoolLayout {
    size_t off_center, off_normal, off_scale, off_nimg, off_images, in_bytes;
    size_t off_ok, off_color, off_ncc, off_fmin, off_x, off_result, off_nevals, off_stage, off_ngrabs, total;
};
PoolLayout pool_layout(size_t n, size_t m) {
    PoolLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
    L.off_center = take(16 * n); L.off_normal = take(16 * n); L.off_scale = take(4 * n); L.off_nimg = take(4 * n);
    L.off_images = take(4 * n * m);
    L.in_bytes = o;
    L.off_ok = take(n); L.off_color = take(12 * n); L.off_ncc = take(4 * n); L.off_fmin = take(8 * n); L.off_x = take(24 * n);
    L.off_result = take(4 * n); L.off_nevals = take(4 * n); L.off_stage = take(4 * n); L.off_ngrabs = take(4 * n);
    L.total = o;
    return L;
}
}  // namespace

// ------------------------------------------------------------------------------------------
// OPEN BATCH: the service behind single-patch host calls.
//
// The reference's callers hand over ONE patch per optimize() call from many host threads (Scene.cpp:166-167,
// CellProcessor.cpp:129,256).  A launch per call costs a kernel start and two PCIe copies; a launch per group of
// callers (the combiner of round 2) makes everybody wait for the slowest patch of the group (450 evaluations against a
// mean of 88).  Here ONE long-running launch serves them all: its patch arrays live in host-mapped memory and have
// room for kCap patches; a calling thread appends its patch, publishes the new count (ctrl[0]) and spins on its own
// done flag, which the kernel sets as soon as that patch's results are visible to the host -- so a caller waits for
// its own patch only.  The launch is closed (ctrl[1] = 1, the kernel drains and exits) when nobody is waiting any
// more, when its room is used up, or when the options change; a second Service object takes over meanwhile.
// Results are those of any other launch: patches are independent.
// HPMVS_SERVICE_STATS=1: where the time of a call goes (printed at exit)
static std::atomic<long long> g_svc_ns_hold{0}, g_svc_ns_acq{0}, g_svc_slow{0}, g_svc_acq_long{0}, g_svc_ns_acq_long{0};
static std::atomic<long long> g_svc_calls{0}, g_svc_ns_lock{0}, g_svc_ns_wait{0}, g_svc_ns_total{0}, g_svc_opens{0};
static bool service_stats() {
    static const bool on = [] {
        const char* e = getenv("HPMVS_SERVICE_STATS");
        const bool v = e && e[0] == '1';
        if (v) atexit([] {
            const double c = (double)std::max<long long>(1, g_svc_calls.load());
            fprintf(stderr, "hpmvs service: %lld calls, %lld launches; per call: submit (lock + copy) %.1f us of which the lock is held %.1f us [to get the submission lock %.1f us on average; %lld calls waited > 200 us for it, %.0f us each: behind a thread that was starting a launch; %lld launches started], wait for the device %.1f us, total %.1f us\n",
                    g_svc_calls.load(), g_svc_opens.load(), g_svc_ns_lock.load() / c / 1e3, g_svc_ns_hold.load() / c / 1e3, g_svc_ns_acq.load() / c / 1e3, g_svc_acq_long.load(), g_svc_ns_acq_long.load() / 1e3 / (double)std::max<long long>(1, g_svc_acq_long.load()), g_svc_slow.load(), g_svc_ns_wait.load() / c / 1e3, g_svc_ns_total.load() / c / 1e3);
        });
        return v;
    }();
    return on;
}

struct hpmvs_scene::Service {
    static constexpr int kCap = 16384;      // patches per open batch
    static constexpr int kLanes = 2;        // patches per wavefront (few: latency first)
    hipStream_t st = nullptr;
    char* host = nullptr;                   // host-mapped, coherent: IN records, OUT records, ctrl words, done flags
    int32_t* ws = nullptr;                  // launch workspace + the slots' staging records
    size_t ws_bytes = 0, stage_off = 0;
    int wgs = 0;
    uint32_t* in = nullptr;                 // kCap x kSvcInDwords
    uint32_t* out = nullptr;                // kCap x kSvcOutDwords
    volatile int32_t* ctrl = nullptr;
    volatile int32_t* done = nullptr;
    int published = 0;
    unsigned gen = 0;                       // launch generation (service_open increments it): a closer ends only the launch it published into
    std::atomic<int> outstanding{0};        // published patches whose owner has not picked up the results yet
    bool open = false, running = false;
    hpmvs_options opts;
};

static void service_close(hpmvs_scene::Service& S) {  // call with svc_mu held
    if (S.open) {
        std::atomic_thread_fence(std::memory_order_release);
        S.ctrl[1] = 1;
        S.open = false;
    }
}
static int service_wait_drained(hpmvs_scene::Service& S) {
    if (S.running) {
        HIPCHK(hipStreamSynchronize(S.st));
        S.running = false;
    }
    return HPMVS_OK;
}
static void service_close_all(const hpmvs_scene* s) {
    std::lock_guard<std::mutex> lk(s->svc_mu);
    for (int k = 0; k < 2; k++)
        if (s->svc[k]) service_close(*s->svc[k]);
}
// closes and drains every open batch of the scene (before anything that synchronises the whole device)
static int service_quiesce(const hpmvs_scene* s) {
    std::lock_guard<std::mutex> lk(s->svc_mu);
    for (int k = 0; k < 2; k++)
        if (s->svc[k]) {
            service_close(*s->svc[k]);
            int rc = service_wait_drained(*s->svc[k]);
            if (rc) return rc;
        }
    return HPMVS_OK;
}
// stream, host-mapped record area and launch workspace of one open-batch object (~12 MB pinned + the workspace: ~12 ms)
static int service_alloc(const hpmvs_scene* s, hpmvs_scene::Service& S) {
    using Service = hpmvs_scene::Service;
    if (S.st) return HPMVS_OK;
    const size_t in_bytes = sizeof(uint32_t) * kSvcInDwords * (size_t)Service::kCap;
    const size_t out_bytes = sizeof(uint32_t) * kSvcOutDwords * (size_t)Service::kCap;
    {
        S.wgs = s->n_cus;  // one workgroup per CU: batch launches keep three quarters of every CU
        if (const char* e = getenv("HPMVS_SERVICE_WGS")) S.wgs = atoi(e) > 0 ? atoi(e) : S.wgs;  // (experiments)
        S.stage_off = (optimize_workspace_bytes_for(S.wgs) + 255) & ~(size_t)255;
        S.ws_bytes = S.stage_off + optimize_stage_bytes_for(S.wgs);
        hipStream_t st = nullptr; char* hv = nullptr; int32_t* wv = nullptr;
        const size_t hbytes = in_bytes + out_bytes + 256 + sizeof(int32_t) * (size_t)Service::kCap;
        bool good = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        good = good && hipHostMalloc((void**)&hv, hbytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
        good = good && hipMalloc((void**)&wv, S.ws_bytes) == hipSuccess;
        if (!good) {
            if (wv) hipFree(wv);
            if (hv) hipHostFree(hv);
            if (st) hipStreamDestroy(st);
            return fail(HPMVS_ERR_HIP, "cannot allocate the open batch of single-patch calls");
        }
        S.st = st; S.host = hv; S.ws = wv;
        S.in = (uint32_t*)hv; S.out = (uint32_t*)(hv + in_bytes);
        S.ctrl = (volatile int32_t*)(hv + in_bytes + out_bytes);
        S.done = (volatile int32_t*)(hv + in_bytes + out_bytes + 256);
    }
    return HPMVS_OK;
}
static int service_open(const hpmvs_scene* s, hpmvs_scene::Service& S, const hpmvs_options* o) {  // s->mu and svc_mu held
    using Service = hpmvs_scene::Service;
    int rc = service_wait_drained(S);
    if (rc) return rc;
    // the owners of the previous generation's patches read their results from S.host: not before they are done
    while (S.outstanding.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    if ((rc = service_alloc(s, S))) return rc;
    memset((void*)S.ctrl, 0, 256 + sizeof(int32_t) * (size_t)Service::kCap);  // ctrl + done flags
    S.opts = *o;
    S.published = 0; S.outstanding.store(0);
    S.gen++;
    DevBatch D;
    memset(&D, 0, sizeof(D));
    D.n = Service::kCap; D.max_images = HPMVS_FAST_IMAGES;   // (the open batch's records hold 64 ids)
    // (unified addressing: the host pointer is the device pointer of mapped memory)
    D.svc_in = (const unsigned long long*)S.in; D.svc_out = S.out; D.svc_stage = (uint32_t*)((char*)S.ws + S.stage_off);
    D.svc_ctrl = (const int32_t*)S.ctrl; D.svc_done = (int32_t*)S.done; D.svc_lanes = Service::kLanes;
    if (const char* e = getenv("HPMVS_SERVICE_LANES")) D.svc_lanes = atoi(e) > 0 ? atoi(e) : D.svc_lanes;  // (experiments)
    HIPCHK(hipMemsetAsync(S.ws, 0, kQueueSlotBytes, S.st));
    // behind every batch launch enqueued so far (they need whole CUs and must not find this launch in their way; launches
    // enqueued later end it: enqueue_refinement)
    for (int k = 0; k < kQueueSlots; k++)
        if (s->slot_used[k]) HIPCHK(hipStreamWaitEvent(S.st, s->slot_done[k], 0));
    launch_optimize_service(dev_scene(s), make_dev_options(o), D, S.ws, S.wgs, S.st);
    HIPCHK(hipGetLastError());
    S.open = true; S.running = true;
    g_svc_opens++;
    return HPMVS_OK;
}

static void service_free(hpmvs_scene* s) {
    service_quiesce(s);
    for (int k = 0; k < 2; k++)
        if (s->svc[k]) {
            if (s->svc[k]->ws) hipFree(s->svc[k]->ws);
            if (s->svc[k]->host) hipHostFree(s->svc[k]->host);
            if (s->svc[k]->st) hipStreamDestroy(s->svc[k]->st);
            delete s->svc[k];
            s->svc[k] = nullptr;
        }
}

static bool service_enabled() {
    static const bool on = [] { const char* e = getenv("HPMVS_SERVICE"); return !(e && e[0] == '0'); }();
    return on;
}

// hpmvs_optimize_batch for a handful of host-resident patches (the unchanged callers' optimize() = batch of one)
static int optimize_through_service(const hpmvs_scene* s, const hpmvs_options* o, hpmvs_patch_batch* b) {
    using Service = hpmvs_scene::Service;
    const int n = b->n, m = b->max_images;
    const bool stats = service_stats();
    const auto t_in = std::chrono::steady_clock::now();
    Service* S = nullptr;
    int k0 = 0;
    unsigned my_gen = 0;
    auto usable = [&](const Service* cur) {
        return cur && cur->open && cur->published + n <= Service::kCap && memcmp(&cur->opts, o, sizeof(*o)) == 0;
    };
    for (;;) {
        {
            // the critical section is a few microseconds; a contended futex hand-over costs far more than that (256
            // callers: ~1 ms per submission in the convoy), so try for a moment before queueing up
            bool got = false;
            for (int spin = 0; spin < 400 && !(got = s->svc_mu.try_lock()); spin++) __builtin_ia32_pause();
            if (!got) s->svc_mu.lock();
        }
        if (usable(s->svc[0] ? s->svc[s->svc_cur] : nullptr)) break;  // (svc_mu stays locked)
        s->svc_mu.unlock();
        g_svc_slow++;
        // No open batch that takes this patch: start one -- under the scene lock first (lock order: mu, then svc_mu), so
        // that it is ordered against batch launches: one enqueued before waits on the stream, one enqueued after ends it.
        std::lock_guard<std::recursive_mutex> l1(s->mu);
        std::lock_guard<std::mutex> l2(s->svc_mu);
        if (!s->svc[0]) {
            // both objects get their memory NOW: the first caller that finds the current batch closed (full, other options, ended by
            // a batch launch) would otherwise stall every other caller for the ~12 ms the second object's allocations take
            // (measured: 257 of the first calls of a 256-thread loop waited 14.7 ms each, profiles/r05/callers_host_side.txt)
            s->svc[0] = new Service(); s->svc[1] = new Service();
            int rc0 = service_alloc(s, *s->svc[0]);
            if (!rc0) rc0 = service_alloc(s, *s->svc[1]);
            if (rc0) return rc0;
        }
        Service* cur = s->svc[s->svc_cur];
        if (usable(cur)) continue;  // somebody else did meanwhile
        service_close(*cur);
        s->svc_cur ^= 1;
        cur = s->svc[s->svc_cur];
        service_close(*cur);  // (can only be open if the options flipped twice)
        const int rc = service_open(s, *cur, o);
        if (rc) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(s->svc_mu, std::adopt_lock);
        const auto t_held = std::chrono::steady_clock::now();
        if (stats) {
            const long long a = std::chrono::duration_cast<std::chrono::nanoseconds>(t_held - t_in).count();
            g_svc_ns_acq += a;
            if (a > 200000) { g_svc_acq_long++; g_svc_ns_acq_long += a; }  // (waited behind somebody who was opening a launch)
        }
        struct Hold { bool on; std::chrono::steady_clock::time_point t0; ~Hold() { if (on) g_svc_ns_hold += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } hold{stats, t_held};
        Service* cur = s->svc[s->svc_cur];
        S = cur;
        my_gen = S->gen;
        k0 = S->published;
        for (int i = 0; i < n; i++) {
            uint32_t* rec = S->in + (size_t)(k0 + i) * kSvcInDwords;
            memcpy(rec + 0, b->center + 4 * i, 12); memcpy(rec + 3, b->scale + i, 4);
            memcpy(rec + 4, b->normal + 4 * i, 12);
            // a list longer than the caller's row cannot come back: failed like every other launch does (stage 100)
            const int32_t ni = b->n_images[i] > m ? (int32_t)(HPMVS_FAST_IMAGES + 1) : b->n_images[i];
            memcpy(rec + 7, &ni, 4);
            memcpy(rec + 8, b->images + (size_t)i * m, 4 * (size_t)m);
        }
        std::atomic_thread_fence(std::memory_order_release);
        S->ctrl[0] = k0 + n;  // published
        S->published += n; S->outstanding.fetch_add(n);
    }
    const auto t_sub = std::chrono::steady_clock::now();
    static const long poll_us = [] { const char* e = getenv("HPMVS_SERVICE_POLL_US"); return e ? atol(e) : 50L; }();
    // wait for my own patches only
    int rc = HPMVS_OK;
    for (int i = 0; i < n && rc == HPMVS_OK; i++) {
        const int k = k0 + i;
        unsigned spins = 0;
        auto t_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
        // a patch is at most 1000 evaluation rounds of well under a millisecond; far beyond that the device is not
        // making progress and the caller gets an error instead of waiting for ever
        static const long limit_s = [] { const char* e = getenv("HPMVS_SERVICE_TIMEOUT_S"); return e && atol(e) > 0 ? atol(e) : 120L; }();
        const auto t_limit = std::chrono::steady_clock::now() + std::chrono::seconds(limit_s);
        while (S->done[k] == 0) {
            // Poll gently: the flags live in host memory that the GPU writes over PCIe, and hundreds of cores reading
            // those lines back to back delay exactly those writes and the device's reads of the progress words
            // (256 threads spinning: 6.7 k patches/s; sleeping 50 us between polls: 50 k).  A patch
            // needs milliseconds on the device, so the sleep costs nothing.
            ++spins;
            if (poll_us > 0) { struct timespec ts = {0, poll_us * 1000L}; nanosleep(&ts, nullptr); }
            else if (spins < 2000) { __builtin_ia32_pause(); continue; }
            else std::this_thread::yield();
            if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() > t_check) {
                // the launch must still be running while my flag is down
                const hipError_t q = hipStreamQuery(S->st);
                if (q != hipErrorNotReady && S->done[k] == 0) { rc = fail(HPMVS_ERR_HIP, "open batch ended without finishing a published patch"); break; }
                if (std::chrono::steady_clock::now() > t_limit) { rc = fail(HPMVS_ERR_HIP, "open batch: a published patch was not finished within " + std::to_string(limit_s) + " s (HPMVS_SERVICE_TIMEOUT_S)"); break; }
                t_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (rc) break;
        if (stats && i == n - 1) g_svc_ns_wait += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_sub).count();
        const uint32_t* rec = S->out + (size_t)k * kSvcOutDwords;
        uint8_t okv = rec[0] ? 1 : 0;
        int32_t nout = (int32_t)rec[5];
        int32_t stg = (int32_t)rec[1];
        if (okv && nout > m) { okv = 0; stg = 100; }  // refined list longer than the caller's row
        b->ok[i] = okv;
        if (okv) {
            memcpy(b->center + 4 * i, rec + 8, 16);
            memcpy(b->normal + 4 * i, rec + 12, 16);
            b->n_images[i] = nout;
            for (int q = 0; q < m; q++) b->images[(size_t)i * m + q] = q < nout ? (int32_t)rec[32 + q] : b->images[(size_t)i * m + q];
            if (b->color) memcpy(b->color + 3 * i, rec + 16, 12);
            if (b->ncc) memcpy(b->ncc + i, rec + 19, 4);
        }
        if (b->fmin) memcpy(b->fmin + i, rec + 20, 8);
        if (b->x) memcpy(b->x + 3 * i, rec + 22, 24);
        if (b->result) b->result[i] = (int32_t)rec[2];
        if (b->nevals) b->nevals[i] = (int32_t)rec[3];
        if (b->stage) b->stage[i] = stg;
        if (b->ngrabs) b->ngrabs[i] = (int32_t)rec[4];
    }
    if (stats) {
        const auto t_out = std::chrono::steady_clock::now();
        g_svc_calls++;
        g_svc_ns_lock += std::chrono::duration_cast<std::chrono::nanoseconds>(t_sub - t_in).count();
        g_svc_ns_total += std::chrono::duration_cast<std::chrono::nanoseconds>(t_out - t_in).count();
    }
    if (S->outstanding.fetch_sub(n, std::memory_order_acq_rel) == n) {  // I was the last one waiting
        std::lock_guard<std::mutex> lk(s->svc_mu);
        // nobody is waiting: let the launch end -- the one I published into (the object may have been reopened since:
        // a thread spinning in service_open goes on as soon as the count reaches zero)
        if (S->gen == my_gen && S->outstanding.load() == 0) service_close(*S);
    }
    return rc;
}

static int optimize_small_host_batch(const hpmvs_scene* s, const hpmvs_options* o, hpmvs_patch_batch* b) {
    const size_t n = (size_t)b->n, m = (size_t)b->max_images;
    const PoolLayout L = pool_layout(n, m);
    hpmvs_scene::StageEntry& e = s->stage[s->stage_next.fetch_add(1) % hpmvs_scene::kStageEntries];
    std::lock_guard<std::mutex> use(e.mu);
    if (!e.st || !e.dev || !e.host) {
        // all three or nothing: a failed allocation must not leave a half-initialised entry behind
        const PoolLayout cap = pool_layout((size_t)hpmvs_scene::kStageCapPatches, (size_t)HPMVS_MAX_IMAGES);
        Exclusive ex(s);  // (allocations wait for an idle device)
        hipStream_t st = e.st;
        char *dv = nullptr, *hv = nullptr;
        bool good = st || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        good = good && hipMalloc((void**)&dv, cap.total) == hipSuccess;
        good = good && hipHostMalloc((void**)&hv, cap.total) == hipSuccess;
        if (!good) {
            if (dv) hipFree(dv);
            if (hv) hipHostFree(hv);
            if (st && !e.st) hipStreamDestroy(st);
            return fail(HPMVS_ERR_HIP, "optimize_batch: cannot allocate the staging block of a small host batch");
        }
        e.st = st; e.dev = dv; e.host = hv;
    }
    char* h = e.host;
    memcpy(h + L.off_center, b->center, 16 * n); memcpy(h + L.off_normal, b->normal, 16 * n);
    memcpy(h + L.off_scale, b->scale, 4 * n); memcpy(h + L.off_nimg, b->n_images, 4 * n);
    memcpy(h + L.off_images, b->images, 4 * n * m);
    HIPCHK(hipMemcpyAsync(e.dev, h, L.in_bytes, hipMemcpyHostToDevice, e.st));
    HIPCHK(hipMemsetAsync(e.dev + L.in_bytes, 0, L.total - L.in_bytes, e.st));
    DevBatch D;
    memset(&D, 0, sizeof(D));
    D.n = b->n; D.max_images = b->max_images;
    char* d = e.dev;
    D.center = (float*)(d + L.off_center); D.normal = (float*)(d + L.off_normal); D.scale = (float*)(d + L.off_scale);
    D.n_images = (int32_t*)(d + L.off_nimg); D.images = (int32_t*)(d + L.off_images);
    D.ok = (uint8_t*)(d + L.off_ok); D.color = (float*)(d + L.off_color); D.ncc = (float*)(d + L.off_ncc);
    D.fmin = (double*)(d + L.off_fmin); D.x = (double*)(d + L.off_x); D.result = (int32_t*)(d + L.off_result);
    D.nevals = (int32_t*)(d + L.off_nevals); D.stage = (int32_t*)(d + L.off_stage); D.ngrabs = (int32_t*)(d + L.off_ngrabs);
    const DevOptions dopt = make_dev_options(o);
    int rc;
    {
        std::lock_guard<std::recursive_mutex> lk(s->mu);
        if ((rc = enqueue_refinement(s, dopt, D, e.st))) return rc;
    }
    HIPCHK(hipMemcpyAsync(h, e.dev, L.total, hipMemcpyDeviceToHost, e.st));  // inputs come back refined in place
    HIPCHK(hipStreamSynchronize(e.st));
    memcpy(b->center, h + L.off_center, 16 * n); memcpy(b->normal, h + L.off_normal, 16 * n);
    memcpy(b->n_images, h + L.off_nimg, 4 * n); memcpy(b->images, h + L.off_images, 4 * n * m);
    memcpy(b->ok, h + L.off_ok, n);
    if (b->color) memcpy(b->color, h + L.off_color, 12 * n);
    if (b->ncc) memcpy(b->ncc, h + L.off_ncc, 4 * n);
    if (b->fmin) memcpy(b->fmin, h + L.off_fmin, 8 * n);
    if (b->x) memcpy(b->x, h + L.off_x, 24 * n);
    if (b->result) memcpy(b->result, h + L.off_result, 4 * n);
    if (b->nevals) memcpy(b->nevals, h + L.off_nevals, 4 * n);
    if (b->stage) memcpy(b->stage, h + L.off_stage, 4 * n);
    if (b->ngrabs) memcpy(b->ngrabs, h + L.off_ngrabs, 4 * n);
    return HPMVS_OK;
}

extern "C" {

int hpmvs_optimize_batch(const hpmvs_scene* s, const hpmvs_options* o, hpmvs_patch_batch* b, int on_device,
                         void* stream) {
    int rc = check_batch(s, o, b);
    if (rc) return rc;
    if (!b->ok) return fail(HPMVS_ERR_ARG, "optimize_batch: ok[] is required");
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    if (!on_device && !stream && b->n <= kServiceMaxPatches && b->max_images <= HPMVS_FAST_IMAGES && service_enabled()) {
        // The open batch runs the 64-id kernel alone.  A patch that comes back with stage 100 (its list outgrew 64 ids, or the
        // caller's row) was left untouched: it goes through a batch launch of its own, whose wide kernel takes lists up to
        // HPMVS_MAX_IMAGES (kernel_wide.hip) -- same result as if it had been part of a batch.
        std::vector<int32_t> stage_tmp;
        hpmvs_patch_batch bb = *b;
        if (!bb.stage) { stage_tmp.assign((size_t)b->n, 0); bb.stage = stage_tmp.data(); }
        if ((rc = optimize_through_service(s, o, &bb))) return rc;
        for (int i = 0; i < b->n; i++) {
            if (bb.ok[i] || bb.stage[i] != 100) continue;
            hpmvs_patch_batch one = bb;
            one.n = 1;
            one.center += 4 * i; one.normal += 4 * i; one.scale += i; one.n_images += i; one.images += (size_t)i * b->max_images; one.ok += i;
            if (one.color) one.color += 3 * i;
            if (one.ncc) one.ncc += i;
            if (one.fmin) one.fmin += i;
            if (one.x) one.x += 3 * i;
            if (one.result) one.result += i;
            if (one.nevals) one.nevals += i;
            if (one.stage) one.stage += i;
            if (one.ngrabs) one.ngrabs += i;
            if ((rc = optimize_small_host_batch(s, o, &one))) return rc;
        }
        return HPMVS_OK;
    }
    if (!on_device && !stream && b->n <= hpmvs_scene::kStageCapPatches) return optimize_small_host_batch(s, o, b);
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    if (!on_device) {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive; ~Staged frees in one of its own)
        rc = stage_batch(sg, b, false, true, true, /*all_written=*/true);
        s->last_staged_bytes = sg.staged_bytes; s->last_mapped_bytes = sg.mapped_bytes;
    } else rc = stage_batch(sg, b, true, true);
    if (rc) return rc;
    if ((rc = stage_flush(sg, st))) return rc;
    const DevOptions d = make_dev_options(o);
    {
        {
            std::lock_guard<std::recursive_mutex> lk(s->mu);  // workspace hand-out + the scene's single event pair
            if ((rc = enqueue_refinement(s, d, sg.d, st))) return rc;
        }
        if (!on_device) {
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    if (!on_device) {
        const size_t n = (size_t)b->n;
        const DevBatch& D = sg.d;
        if ((rc = stage_out(b->center, D.center, 4 * n))) return rc;
        if ((rc = stage_out(b->normal, D.normal, 4 * n))) return rc;
        if ((rc = stage_out(b->n_images, D.n_images, n))) return rc;
        if ((rc = stage_out(b->images, D.images, n * b->max_images))) return rc;
        if ((rc = stage_out(b->ok, D.ok, n))) return rc;
        if ((rc = stage_out(b->color, D.color, 3 * n))) return rc;
        if ((rc = stage_out(b->ncc, D.ncc, n))) return rc;
        if ((rc = stage_out(b->fmin, D.fmin, n))) return rc;
        if ((rc = stage_out(b->x, D.x, 3 * n))) return rc;
        if ((rc = stage_out(b->result, D.result, n))) return rc;
        if ((rc = stage_out(b->nevals, D.nevals, n))) return rc;
        if ((rc = stage_out(b->stage, D.stage, n))) return rc;
        if ((rc = stage_out(b->ngrabs, D.ngrabs, n))) return rc;
    }
    return HPMVS_OK;
}

int hpmvs_init_patches_batch(const hpmvs_scene* s, const hpmvs_options* o, int start_level, int n_points,
                             const double* xyz, const int32_t* meas_off, const int32_t* meas_img, hpmvs_patch_batch* b,
                             int on_device, void* stream) {
    int rc = check_batch_shape(s, o, b);
    if (rc) return rc;
    if (b->n != n_points) return fail(HPMVS_ERR_ARG, "init_patches_batch: b->n must equal n_points");
    if (n_points > 0 && (!xyz || !meas_off || !b->center || !b->normal || !b->scale || !b->n_images ||
                         !b->images || !b->ok))
        return fail(HPMVS_ERR_ARG, "init_patches_batch: missing array");
    // meas_img may be NULL when no point has a measurement (every seed then fails with stage 10, like the
    // reference's empty image list); with host arrays that is checkable here
    if (n_points > 0 && !meas_img && (on_device || meas_off[n_points] != 0))
        return fail(HPMVS_ERR_ARG, "init_patches_batch: meas_img missing");
    if (start_level < 0 || start_level > o->MAXLEVEL) return fail(HPMVS_ERR_ARG, "init_patches_batch: bad start_level");
    if (n_points == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    const double* dxyz = xyz;
    const int32_t *doff = meas_off, *dimg = meas_img;
    const size_t n = (size_t)n_points;
    if (!on_device) {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, true, /*copy_inputs=*/false))) return rc;
        double* tx = nullptr; int32_t *to = nullptr, *ti = nullptr;
        const size_t nm = (size_t)meas_off[n_points];
        if ((rc = stage_in(sg, tx, xyz, 3 * n, true))) return rc;
        if ((rc = stage_in(sg, to, meas_off, n + 1, true))) return rc;
        if ((rc = stage_in(sg, ti, meas_img, nm ? nm : 1, nm != 0))) return rc;
        dxyz = tx; doff = to; dimg = ti;
    } else if ((rc = stage_batch(sg, b, true, true, /*copy_inputs=*/false))) return rc;
    if ((rc = stage_flush(sg))) return rc;
    const DevOptions d = make_dev_options(o);
    {
        {
            std::lock_guard<std::recursive_mutex> lk(s->mu);
            launch_seed_init(dev_scene(s), d, start_level, n_points, dxyz, doff, dimg, sg.d, st);
            HIPCHK(hipGetLastError());
            if ((rc = enqueue_refinement(s, d, sg.d, st))) return rc;
            launch_drift_gate(n_points, dxyz, sg.d, st);
            HIPCHK(hipGetLastError());
        }
        if (!on_device) {
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    if (!on_device) {
        const DevBatch& D = sg.d;
        if ((rc = stage_out(b->center, D.center, 4 * n))) return rc;
        if ((rc = stage_out(b->normal, D.normal, 4 * n))) return rc;
        if ((rc = stage_out(b->scale, D.scale, n))) return rc;
        if ((rc = copy_outputs(b, D))) return rc;
    }
    return HPMVS_OK;
}

int hpmvs_expand_directions(int mode, float* dxdy) {
    if ((mode != HPMVS_EXPAND_EXTEND && mode != HPMVS_EXPAND_BRANCH) || !dxdy) return fail(HPMVS_ERR_ARG, "expand_directions: bad mode/pointer");
    const int N = expand_fanout(mode);
    for (int k = 0; k < 2 * N; k++) dxdy[k] = expand_direction_table(mode)[k];
    return N;
}

int hpmvs_expand_batch(const hpmvs_scene* s, const hpmvs_options* o, int mode, const hpmvs_patch_batch* parents,
                       const float* cell_center, const float* cell_width, const uint8_t* skip, hpmvs_patch_batch* out,
                       int on_device, void* stream) {
    if (mode != HPMVS_EXPAND_EXTEND && mode != HPMVS_EXPAND_BRANCH) return fail(HPMVS_ERR_ARG, "expand_batch: bad mode");
    int rc = check_batch(s, o, parents);
    if (rc) return rc;
    if ((rc = check_batch_shape(s, o, out))) return rc;
    const int N = expand_fanout(mode);
    if (out->n != parents->n * N) return fail(HPMVS_ERR_ARG, "expand_batch: out->n must be parents->n * fan-out");
    if (out->max_images != parents->max_images) return fail(HPMVS_ERR_ARG, "expand_batch: max_images mismatch");
    if (parents->n > 0 && (!cell_center || !cell_width || !out->center || !out->normal || !out->scale || !out->n_images ||
                           !out->images || !out->ok))
        return fail(HPMVS_ERR_ARG, "expand_batch: missing array");
    if (parents->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sp, sg;
    sp.scene = s; sg.scene = s;  // (this entry point had been left out of the Exclusive protocol)
    const size_t n = (size_t)parents->n;
    const float *dcc = cell_center, *dcw = cell_width;
    const uint8_t* dskip = skip;
    if (!on_device) {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sp, parents, false, false))) return rc;
        if ((rc = stage_batch(sg, out, false, true, /*copy_inputs=*/false))) return rc;
        float *tc = nullptr, *tw = nullptr;
        uint8_t* ts = nullptr;
        if ((rc = stage_in(sp, tc, cell_center, 3 * n, true))) return rc;
        if ((rc = stage_in(sp, tw, cell_width, n, true))) return rc;
        if ((rc = stage_in(sp, ts, skip, n * N, true))) return rc;
        dcc = tc; dcw = tw; dskip = ts;
    } else {
        if ((rc = stage_batch(sp, parents, true, false))) return rc;
        if ((rc = stage_batch(sg, out, true, true, /*copy_inputs=*/false))) return rc;
    }
    if ((rc = stage_flush(sp)) || (rc = stage_flush(sg))) return rc;
    const DevOptions d = make_dev_options(o);
    {
        std::lock_guard<std::recursive_mutex> lk(s->mu);
        launch_expand_init(dev_scene(s), mode, parents->n, sp.d, dcc, dcw, dskip, sg.d, st);
        HIPCHK(hipGetLastError());
        if ((rc = enqueue_refinement(s, d, sg.d, st))) return rc;
        launch_expand_gate(mode, parents->n, sp.d, dcc, dcw, sg.d, st);
        HIPCHK(hipGetLastError());
    }
    if (!on_device) {
        HIPCHK(hipStreamSynchronize(st));
        const DevBatch& D = sg.d;
        const size_t m = (size_t)out->n;
        if ((rc = stage_out(out->center, D.center, 4 * m))) return rc;
        if ((rc = stage_out(out->normal, D.normal, 4 * m))) return rc;
        if ((rc = stage_out(out->scale, D.scale, m))) return rc;
        if ((rc = copy_outputs(out, D))) return rc;
    }
    return HPMVS_OK;
}

// hpmvs_objective_batch is answered by the refinement kernel's own evaluator driven stand-alone (kernels_basic.hip).  A
// measurement build (`make LANE=1`, tools/objective_ab.py) also carries the one-lane-per-grab kernel of
// kernel_objective_lane.hip and runs it with HPMVS_OBJECTIVE_KERNEL=lane: same values, no faster (profiles/r05/lane_per_grab.txt).
static void launch_objective_chosen(const DevScene& sc, const DevOptions& d, const DevBatch& b, const double* x, double* f,
                                    int32_t* g, hipStream_t st) {
#ifdef HPMVS_WITH_LANE_KERNEL
    const char* e = getenv("HPMVS_OBJECTIVE_KERNEL");
    if (e && strcmp(e, "lane") == 0) { launch_objective_lane(sc, d, b, x, f, g, st); return; }
#endif
    launch_objective(sc, d, b, x, f, g, st);
}

int hpmvs_objective_batch(const hpmvs_scene* s, const hpmvs_options* o, const hpmvs_patch_batch* b, const double* x,
                          double* f_out, int32_t* ngrabs_out, int on_device, void* stream) {
    int rc = check_batch(s, o, b);
    if (rc) return rc;
    if (!x || !f_out) return fail(HPMVS_ERR_ARG, "objective_batch: x/f_out missing");
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    const DevOptions d = make_dev_options(o);
    const size_t n = (size_t)b->n;
    if (on_device) {
        if ((rc = stage_batch(sg, b, true, false))) return rc;
        launch_objective_chosen(dev_scene(s), d, sg.d, x, f_out, ngrabs_out, st);
        HIPCHK(hipGetLastError());
        return HPMVS_OK;
    }
    double *dx = nullptr, *df = nullptr;
    int32_t* dg = nullptr;
    {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if ((rc = stage_in(sg, dx, x, 3 * n, true))) return rc;
        if ((rc = stage_in(sg, df, f_out, n, false))) return rc;
        if ((rc = stage_in(sg, dg, ngrabs_out, n, false))) return rc;
    }
    if ((rc = stage_flush(sg))) return rc;
    launch_objective_chosen(dev_scene(s), d, sg.d, dx, df, dg, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if ((rc = stage_out(f_out, df, n))) return rc;
    if ((rc = stage_out(ngrabs_out, dg, n))) return rc;
    return HPMVS_OK;
}

int hpmvs_inccs_batch(const hpmvs_scene* s, const hpmvs_options* o, const hpmvs_patch_batch* b, int ref_idx, int robust,
                      float* out, int on_device, void* stream) {
    int rc = check_batch(s, o, b);
    if (rc) return rc;
    if (!out || ref_idx < 0) return fail(HPMVS_ERR_ARG, "inccs_batch: out/ref_idx");
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    const DevOptions d = make_dev_options(o);
    const size_t n = (size_t)b->n;
    if (on_device) {
        if ((rc = stage_batch(sg, b, true, false))) return rc;
        launch_inccs(dev_scene(s), d, sg.d, ref_idx, robust, out, st);
        HIPCHK(hipGetLastError());
        return HPMVS_OK;
    }
    float* dout = nullptr;
    {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if ((rc = stage_in(sg, dout, out, n * b->max_images, false))) return rc;
    }
    if ((rc = stage_flush(sg))) return rc;
    launch_inccs(dev_scene(s), d, sg.d, ref_idx, robust, dout, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return stage_out(out, dout, n * b->max_images);
}

// ---- depth maps and the acceptance gates over them (kernel_depth.hip)
int hpmvs_scene_depth_reset(hpmvs_scene* s) {
    if (!s || !s->committed) return fail(HPMVS_ERR_STATE, "scene_depth_reset: scene not committed");
    HIPCHK(hipSetDevice(s->device));
    Exclusive ex(s);  // (the first call allocates: hipMalloc waits for an idle device, see Exclusive)
    if (!s->depth_pool || !s->ddepth) {
        // Scene.cpp:74-80: rows = height(level) / DEPTH_SUBSAMPLE, cols = width(level) / DEPTH_SUBSAMPLE (a double 2)
        size_t total = 0;
        std::vector<DevDepthView> hd((size_t)s->n_views, DevDepthView{});
        for (int v = 0; v < s->n_views; v++)
            for (int l = 0; l < s->hviews[v].n_levels; l++) {
                hd[v].rows[l] = (int)((double)s->hviews[v].h[l] / 2.0);
                hd[v].cols[l] = (int)((double)s->hviews[v].w[l] / 2.0);
                total += (size_t)hd[v].rows[l] * hd[v].cols[l];
            }
        if (total == 0) return fail(HPMVS_ERR_STATE, "scene_depth_reset: empty scene");
        // both buffers into locals; the scene sees them only when everything succeeded (a failure half way must not
        // leave a pool without its table: the kernels test one and dereference the other)
        float* pool = nullptr;
        DevDepthView* table = nullptr;
        struct Undo { float*& a; DevDepthView*& b; ~Undo() { if (a) hipFree(a); if (b) hipFree(b); } } undo{pool, table};
        HIPCHK(hipMalloc((void**)&pool, total * sizeof(float)));
        size_t off = 0;
        for (int v = 0; v < s->n_views; v++)
            for (int l = 0; l < s->hviews[v].n_levels; l++) {
                hd[v].d[l] = pool + off;
                off += (size_t)hd[v].rows[l] * hd[v].cols[l];
            }
        HIPCHK(hipMalloc((void**)&table, sizeof(DevDepthView) * (size_t)s->n_views));
        HIPCHK(hipMemcpy(table, hd.data(), sizeof(DevDepthView) * (size_t)s->n_views, hipMemcpyHostToDevice));
        if (s->depth_pool) hipFree(s->depth_pool);  // (a half-initialised scene of an older library state)
        if (s->ddepth) hipFree(s->ddepth);
        s->depth_pool = pool; s->ddepth = table; s->depth_floats = total; s->hdepth.swap(hd);
        s->bytes += total * sizeof(float) + sizeof(DevDepthView) * (size_t)s->n_views;
        pool = nullptr; table = nullptr;
    }
    launch_depth_fill(s->depth_pool, s->depth_floats, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));  // (not the whole device: an open batch of single-patch calls may be running)
    return HPMVS_OK;
}

static int depth_level(const hpmvs_scene* s, int view, int level, const char* who, float** p, int* rows, int* cols) {
    if (!s || !s->depth_pool || !s->ddepth) return fail(HPMVS_ERR_STATE, std::string(who) + ": call hpmvs_scene_depth_reset first");
    if (view < 0 || view >= s->n_views || level < 0 || level >= s->hviews[view].n_levels)
        return fail(HPMVS_ERR_ARG, std::string(who) + ": bad view / level");
    *p = s->hdepth[view].d[level]; *rows = s->hdepth[view].rows[level]; *cols = s->hdepth[view].cols[level];
    return HPMVS_OK;
}
int hpmvs_scene_depth_set_level(hpmvs_scene* s, int view, int level, const float* data, int rows, int cols) {
    float* p; int r, c, rc;
    if ((rc = depth_level(s, view, level, "scene_depth_set_level", &p, &r, &c))) return rc;
    if (!data || rows != r || cols != c) return fail(HPMVS_ERR_ARG, "scene_depth_set_level: shape is not rows x cols of that level");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpy(p, data, sizeof(float) * (size_t)r * c, hipMemcpyHostToDevice));
    return HPMVS_OK;
}
int hpmvs_scene_depth_get_level(const hpmvs_scene* s, int view, int level, float* data, size_t capacity, int* rows, int* cols) {
    float* p; int r, c, rc;
    if ((rc = depth_level(s, view, level, "scene_depth_get_level", &p, &r, &c))) return rc;
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (!data) return HPMVS_OK;  // shape query
    if (capacity < (size_t)r * c) return fail(HPMVS_ERR_ARG, "scene_depth_get_level: buffer too small");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipMemcpy(data, p, sizeof(float) * (size_t)r * c, hipMemcpyDeviceToHost));
    return HPMVS_OK;
}

static int check_depth_batch(const hpmvs_scene* s, const hpmvs_patch_batch* b, const char* who) {
    if (!s || !b) return fail(HPMVS_ERR_ARG, std::string(who) + ": null scene / batch");
    if (!s->committed || !s->depth_pool || !s->ddepth) return fail(HPMVS_ERR_STATE, std::string(who) + ": call hpmvs_scene_depth_reset first");
    if (b->n < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, std::string(who) + ": bad n/max_images");
    if (b->n > 0 && (!b->center || !b->normal || !b->scale || !b->n_images || !b->images))
        return fail(HPMVS_ERR_ARG, std::string(who) + ": batch input arrays missing");
    return HPMVS_OK;
}
int hpmvs_set_depths_batch(hpmvs_scene* s, const hpmvs_patch_batch* b, int on_device, void* stream) {
    int rc = check_depth_batch(s, b, "set_depths_batch");
    if (rc) return rc;
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    if (!on_device) {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if (b->ok) { if ((rc = stage_in(sg, sg.d.ok, b->ok, (size_t)b->n, true))) return rc; }
    } else if ((rc = stage_batch(sg, b, true, false))) return rc;
    if ((rc = stage_flush(sg))) return rc;
    launch_set_depths(dev_scene(s), s->ddepth, sg.d, st);
    HIPCHK(hipGetLastError());
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return HPMVS_OK;
}
// Scene::setDepths(patch_i, subtract[i]) for i = 0 .. n - 1 IN THAT ORDER (kernel_depth.hip: keys, sort, per-cell replay).
// Without a subtraction the calls commute and the batch goes to set_depths_kernel.
int hpmvs_depth_ops_batch(hpmvs_scene* s, const hpmvs_patch_batch* b, const uint8_t* subtract, int on_device, void* stream) {
    int rc = check_depth_batch(s, b, "depth_ops_batch");
    if (rc) return rc;
    if (b->n == 0) return HPMVS_OK;
    if (!subtract) return hpmvs_set_depths_batch(s, b, on_device, stream);
    const size_t cap = (size_t)b->n * (size_t)b->max_images;
    if (cap >= ((size_t)1 << 28)) return fail(HPMVS_ERR_ARG, "depth_ops_batch: n * max_images must stay below 2^28 (apply the level in pieces: the order between pieces is kept)");
    if (s->depth_floats >= ((size_t)1 << 32)) return fail(HPMVS_ERR_STATE, "depth_ops_batch: the scene's depth maps hold 2^32 cells or more");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    const uint8_t* dsub = nullptr;
    unsigned long long *keys = nullptr, *sorted = nullptr;
    unsigned int* counter = nullptr;
    {
        Exclusive ex(s);  // the allocations of this call (see Exclusive)
        if (!on_device) {
            if ((rc = stage_batch(sg, b, false, false))) return rc;
            if (b->ok) { if ((rc = stage_in(sg, sg.d.ok, b->ok, (size_t)b->n, true))) return rc; }
            uint8_t* t = nullptr;
            if ((rc = stage_in(sg, t, subtract, (size_t)b->n, true))) return rc;
            dsub = t;
        } else {
            if ((rc = stage_batch(sg, b, true, false))) return rc;
            dsub = subtract;
        }
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, cap * sizeof(unsigned long long))); sg.tmp.push_back(p); keys = (unsigned long long*)p;
        HIPCHK(hipMalloc(&p, cap * sizeof(unsigned long long))); sg.tmp.push_back(p); sorted = (unsigned long long*)p;
        HIPCHK(hipMalloc(&p, sizeof(unsigned int))); sg.tmp.push_back(p); counter = (unsigned int*)p;
    }
    if ((rc = stage_flush(sg))) return rc;
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned int), st));
    launch_depth_ops_keys(dev_scene(s), s->ddepth, s->depth_pool, sg.d, keys, counter, st);
    HIPCHK(hipGetLastError());
    unsigned int count = 0;
    HIPCHK(hipMemcpyAsync(&count, counter, sizeof(count), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (count == 0) return HPMVS_OK;
    size_t temp_bytes = 0;
    if (depth_ops_sort(nullptr, &temp_bytes, keys, sorted, count, st) != 0) return fail(HPMVS_ERR_HIP, "depth_ops_batch: rocPRIM size query failed");
    void* temp = nullptr;
    {
        Exclusive ex(s);
        HIPCHK(hipMalloc(&temp, temp_bytes ? temp_bytes : 1)); sg.tmp.push_back(temp);
    }
    if (depth_ops_sort(temp, &temp_bytes, keys, sorted, count, st) != 0) return fail(HPMVS_ERR_HIP, "depth_ops_batch: rocPRIM sort failed");
    launch_depth_ops_apply(dev_scene(s), s->ddepth, s->depth_pool, sg.d, dsub, sorted, count, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));   // (the temporaries are freed when this call returns)
    return HPMVS_OK;
}
// Scene::getLevelSupport(patch, minLevel) (src/hpmvs/Scene.cpp:334-343) per patch
int hpmvs_level_support_batch(const hpmvs_scene* s, const hpmvs_patch_batch* b, int min_level, int32_t* support, int on_device, void* stream) {
    if (!s || !b || !support) return fail(HPMVS_ERR_ARG, "level_support_batch: null argument");
    if (!s->committed) return fail(HPMVS_ERR_STATE, "level_support_batch: scene not committed");
    if (b->n < 0 || b->max_images < 1 || b->max_images > HPMVS_MAX_IMAGES) return fail(HPMVS_ERR_ARG, "level_support_batch: bad n/max_images");
    if (b->n > 0 && (!b->center || !b->scale || !b->n_images || !b->images)) return fail(HPMVS_ERR_ARG, "level_support_batch: batch input arrays missing");
    if (b->n == 0) return HPMVS_OK;
    int rc;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    if (on_device) {
        if ((rc = stage_batch(sg, b, true, false))) return rc;
        launch_level_support(dev_scene(s), sg.d, min_level, support, st);
        HIPCHK(hipGetLastError());
        return HPMVS_OK;
    }
    int32_t* dsup = nullptr;
    {
        Exclusive ex(s);
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if ((rc = stage_in(sg, dsup, support, (size_t)b->n, false))) return rc;
    }
    if ((rc = stage_flush(sg))) return rc;
    launch_level_support(dev_scene(s), sg.d, min_level, dsup, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return stage_out(support, dsup, (size_t)b->n);
}
int hpmvs_depth_gates_batch(const hpmvs_scene* s, const hpmvs_patch_batch* b, float margin, int abs_int,
                            int32_t* n_visible, int32_t* n_blocking, int32_t* n_free, int on_device, void* stream) {
    int rc = check_depth_batch(s, b, "depth_gates_batch");
    if (rc) return rc;
    if (!n_visible && !n_blocking && !n_free) return fail(HPMVS_ERR_ARG, "depth_gates_batch: no output array");
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    if (on_device) {
        if ((rc = stage_batch(sg, b, true, false))) return rc;
        launch_depth_gates(dev_scene(s), s->ddepth, sg.d, margin, abs_int, n_visible, n_blocking, n_free, st);
        HIPCHK(hipGetLastError());
        return HPMVS_OK;
    }
    const size_t n = (size_t)b->n;
    int32_t *dv = nullptr, *db = nullptr, *df = nullptr;
    {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if ((rc = stage_in(sg, dv, n_visible, n, false))) return rc;
        if ((rc = stage_in(sg, db, n_blocking, n, false))) return rc;
        if ((rc = stage_in(sg, df, n_free, n, false))) return rc;
    }
    if ((rc = stage_flush(sg))) return rc;
    launch_depth_gates(dev_scene(s), s->ddepth, sg.d, margin, abs_int, dv, db, df, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if ((rc = stage_out(n_visible, dv, n))) return rc;
    if ((rc = stage_out(n_blocking, db, n))) return rc;
    return stage_out(n_free, df, n);
}

int hpmvs_depth_footprints_batch(const hpmvs_scene* s, const hpmvs_patch_batch* b, int32_t* writes, int32_t* frees,
                                 int32_t* attached, int32_t* view_block, int on_device, void* stream) {
    int rc = check_depth_batch(s, b, "depth_footprints_batch");
    if (rc) return rc;
    if (!writes && !frees && !attached && !view_block) return fail(HPMVS_ERR_ARG, "depth_footprints_batch: no output array");
    if (b->n == 0) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    Staged sg;
    sg.scene = s;
    if (on_device) {
        if ((rc = stage_batch(sg, b, true, false))) return rc;
        launch_depth_footprints(dev_scene(s), s->ddepth, sg.d, writes, frees, attached, view_block, st);
        HIPCHK(hipGetLastError());
        return HPMVS_OK;
    }
    const size_t n = (size_t)b->n, M = (size_t)b->max_images, V = (size_t)s->n_views;
    int32_t *dw = nullptr, *df = nullptr, *da = nullptr, *dv = nullptr;
    {
        Exclusive ex(s);  // the allocations of a host-staged call (see Exclusive)
        if ((rc = stage_batch(sg, b, false, false))) return rc;
        if ((rc = stage_in(sg, dw, writes, n * M * 4, false))) return rc;
        if ((rc = stage_in(sg, df, frees, n * M * 4, false))) return rc;
        if ((rc = stage_in(sg, da, attached, n * M * 3, false))) return rc;
        if ((rc = stage_in(sg, dv, view_block, n * V * 3, false))) return rc;
    }
    if ((rc = stage_flush(sg))) return rc;
    launch_depth_footprints(dev_scene(s), s->ddepth, sg.d, dw, df, da, dv, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if ((rc = stage_out(writes, dw, n * M * 4))) return rc;
    if ((rc = stage_out(frees, df, n * M * 4))) return rc;
    if ((rc = stage_out(attached, da, n * M * 3))) return rc;
    return stage_out(view_block, dv, n * V * 3);
}

// how many patches of the last refinement launch outgrew the batch kernel's 64-id rows and were redone by the wide kernel
int hpmvs_last_wide_patches(const hpmvs_scene* s, int32_t* n) {
    if (!s || !n) return fail(HPMVS_ERR_ARG, "last_wide_patches: null argument");
    *n = 0;
    if (!s->last_ws) return HPMVS_OK;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(n, s->last_ws + 241, sizeof(int32_t), hipMemcpyDeviceToHost));
    return HPMVS_OK;
}

// debug: phase counters of the last optimize launch (only meaningful in -DHPMVS_PROFILE_PHASES builds)
int hpmvs_debug_counters(const hpmvs_scene* s, unsigned long long* out4) {
    if (!s || !out4) return fail(HPMVS_ERR_ARG, "debug_counters: null argument");
    HIPCHK(hipSetDevice(s->device));
    { const int rcq = service_quiesce(s); if (rcq) return rcq; }
    HIPCHK(hipDeviceSynchronize());
    if (!s->last_queue) return fail(HPMVS_ERR_STATE, "debug_counters: no launch yet");
    HIPCHK(hipMemcpy(out4, (char*)s->last_queue + 16, 544, hipMemcpyDeviceToHost));
    return HPMVS_OK;
}

// how the arrays of the last host-pointer hpmvs_optimize_batch call of this scene (the path above the small-batch
// thresholds) reached the device: bytes copied through device buffers, bytes used in place (pinned, mapped host memory)
int hpmvs_last_staging(const hpmvs_scene* s, unsigned long long* staged_bytes, unsigned long long* in_place_bytes) {
    if (!s || !staged_bytes || !in_place_bytes) return fail(HPMVS_ERR_ARG, "last_staging: null argument");
    *staged_bytes = s->last_staged_bytes; *in_place_bytes = s->last_mapped_bytes;
    return HPMVS_OK;
}

// pinned host memory mapped into the GPUs' address space, for callers without the HIP headers: arrays of a host-pointer
// call that live in it are used in place (see mapped_host)
void* hpmvs_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void hpmvs_host_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

int hpmvs_last_kernel_ms(const hpmvs_scene* s, float* ms) {
    if (!s || !ms) return fail(HPMVS_ERR_ARG, "last_kernel_ms: null argument");
    if (!s->timed) return fail(HPMVS_ERR_STATE, "last_kernel_ms: no optimize launch yet");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipEventSynchronize(s->ev1));
    HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return HPMVS_OK;
}

}  // extern "C"
