// kernel_depth.hip -- per-view depth maps in HBM and the acceptance gates that read them (second half).
//
// The reference's expansion accepts a refined candidate only if it is consistent with the depth maps the
// already accepted patches have written (src/hpmvs/CellProcessor.cpp:134-142):
//     scene.depthTests(p, margin)     >= MIN_IMAGES     (src/hpmvs/Scene.cpp:518-524, 531-580)
//     scene.viewBlockTest(p, margin)  <  MIN_IMAGES     (Scene.cpp:607-642: ALL views, 3x3 pixels, 6 pyramid levels each)
//     scene.pixelFreeTests(p)         >= MIN_IMAGES - 1 and > 0.75 * #images   (Scene.cpp:582-605)
// and Scene::setDepths (Scene.cpp:351-381) records an accepted patch.  On the host that is a serial walk over up to
// 200 views per candidate; here a wavefront takes one candidate, lane v one view (chunks of 64), and the three counts
// are ballots.  The maps keep the reference's layout (Eigen::MatrixXf(rows, cols), column-major, MAX_DEPTH = 1000,
// rows/cols = level size / DEPTH_SUBSAMPLE) so that the scheduler's matrices can be copied in and out as they are.
// Read-only on the gate side; setDepths is a float minimum (order-independent), done with atomicMin on the bit
// patterns of non-negative floats.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>

#include "launch.h"
#include "photometric.hpp"

namespace hpmvs {

namespace {

constexpr float kMaxDepth = 1000.0f;  // Scene::MAX_DEPTH (Scene.cpp:33)

// `int / DEPTH_SUBSAMPLE` with DEPTH_SUBSAMPLE a double 2 (Scene.h:78), stored to an int: truncation towards zero
__device__ __forceinline__ int sub_index(int v) { return (int)((double)v / 2.0); }

// Camera::getLevel (Camera.cpp:92-95) then std::round
__device__ __forceinline__ int level_rounded(const DevView& V, const float* c, float scale) {
    const float d0 = c[0] - V.center[0], d1 = c[1] - V.center[1], d2 = c[2] - V.center[2];
    const float fz = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    const float lv = (float)log2((double)(scale * V.fsum) / (2.0 * (double)fz));
    return cvt_i32_x86(roundf(lv));
}
// Camera::mult at `level` (Camera.h:76-78): P[level] = diag(2^-l, 2^-l, 1) * P0 exactly
__device__ __forceinline__ void mult_level(const DevView& V, const float* c, int level, float* r) {
    const float* P = V.P0;
    const float s = __int_as_float((127 - level) << 23);
    r[0] = (((P[0] * c[0] + P[1] * c[1]) + P[2] * c[2]) + P[3]) * s;
    r[1] = (((P[4] * c[0] + P[5] * c[1]) + P[6] * c[2]) + P[7]) * s;
    r[2] = ((P[8] * c[0] + P[9] * c[1]) + P[10] * c[2]) + P[11];
}
// Camera::project at `level` (Camera.h:45-62)
__device__ __forceinline__ void project_level(const DevView& V, const float* c, int level, float* r) {
    float P[12];
    for (int k = 0; k < 12; k++) P[k] = V.P0[k];
    const float z = ((P[8] * c[0] + P[9] * c[1]) + P[10] * c[2]) + P[11];
    if (z <= 0.0f) { r[0] = -65535.0f; r[1] = -65535.0f; r[2] = -1.0f; return; }
    float u, v;
    project0(P, c[0], c[1], c[2], u, v);
    const float s = __int_as_float((127 - level) << 23);
    r[0] = u * s; r[1] = v * s; r[2] = z / z;
}
// Scene::getFullDepth (Scene.cpp:406-432)
__device__ __forceinline__ float full_depth(const DevDepthView& D, int n_levels, int xx, int yy) {
    float depth = kMaxDepth;
    int x = sub_index(xx), y = sub_index(yy);
    for (int level = 0; level < n_levels; level++) {
        if (x < 0 || x >= D.cols[level] || y < 0 || y >= D.rows[level]) return depth;
        const float v = D.d[level][y + (size_t)x * D.rows[level]];
        depth = v < depth ? v : depth;
        x /= 2; y /= 2;
    }
    return depth;
}
// the 3x3 level-0 pixel block Scene::depthTest reads (Scene.cpp:538-550): its top-left pixel and the patch's depth there
__device__ __forceinline__ void depth_test_block(const DevView& V, const float* c, int& ix0, int& iy0, float& depth) {
    float m[3];
    mult_level(V, c, 0, m);
    depth = m[2];
    ix0 = cvt_i32_x86((double)(m[0] / m[2]) + 0.5) - 1; iy0 = cvt_i32_x86((double)(m[1] / m[2]) + 0.5) - 1;
}
// the one cell Scene::pixelFreeTest reads (Scene.cpp:590-604); false: it returns before reading a cell
__device__ __forceinline__ bool pixel_free_cell(const DevView& V, const DevDepthView& D, const float* c, float scale, int& level, int& x, int& y) {
    level = level_rounded(V, c, scale);
    if (!(level >= 0 && level < V.n_levels)) return false;
    float r[3];
    project_level(V, c, level, r);
    const int ix = cvt_i32_x86((double)(r[0] / r[2]) + 0.5), iy = cvt_i32_x86((double)(r[1] / r[2]) + 0.5);
    if (ix < 0 || ix >= V.w[level] || iy < 0 || iy >= V.h[level]) return false;
    x = sub_index(ix); y = sub_index(iy);
    return !(x < 0 || x >= D.cols[level] || y < 0 || y >= D.rows[level]);
}
// the one cell Scene::setDepths writes for an attached image (Scene.cpp:353-372) and the depth it offers; false: none
__device__ __forceinline__ bool set_depths_cell(const DevScene& sc, const DevView& V, const DevDepthView& D, const float* c, float scale,
                                                int& level, int& x, int& y, float& d) {
    const float d0 = c[0] - V.center[0], d1 = c[1] - V.center[1], d2 = c[2] - V.center[2];
    level = level_clamped(sc, scale, V.fsum, sqrtf((d0 * d0 + d1 * d1) + d2 * d2), V.n_levels - 1);
    float m[3];
    mult_level(V, c, level, m);
    x = sub_index(cvt_i32_x86((double)(m[0] / m[2]) + 0.5)); y = sub_index(cvt_i32_x86((double)(m[1] / m[2]) + 0.5));
    d = m[2];
    if (x < 0 || x >= D.cols[level] || y < 0 || y >= D.rows[level]) return false;
    return m[2] >= 0.0f;  // (the reference CHECK-fails on a negative depth)
}
// does Scene::viewBlockTest look at this view at all (Scene.cpp:611-621)?
__device__ __forceinline__ bool view_block_visible(const DevView& V, const float* c, float scale) {
    const int level = level_rounded(V, c, scale);
    if (level < 0 || level > V.n_levels - 1) return false;
    float r[3];
    project_level(V, c, level, r);
    return !(r[0] < 0 || r[0] > (float)V.w[level] || r[1] < 0 || r[1] > (float)V.h[level]);
}
// Scene::depthTest(patch, image, margin, neighbours = true, viewBlock) (Scene.cpp:531-580)
__device__ __forceinline__ bool depth_test(const DevView& V, const DevDepthView& D, const float* c, const float* n,
                                           float scale, float margin, bool view_block, int abs_int) {
    int ix0, iy0;
    float depth;
    depth_test_block(V, c, ix0, iy0, depth);
    // patch-constant part of the pixel test
    float ray[3];
    const float d[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
    normalized3f(d, ray);
    const float rn = dot3f(ray, n) + 0.0f;  // the 4-vector dot: ... + w * w with both w = 0
    const float factor = 2.0f < 2.0f + rn ? 2.0f : 2.0f + rn;
    const double thr = (double)(scale * margin * factor) * 2.0;
    for (int yy = 0; yy < 3; yy++)
        for (int xx = 0; xx < 3; xx++) {
            const int ix = ix0 + xx, iy = iy0 + yy;
            if (depth < 0 || ix < 0 || ix >= V.w[0] || iy < 0 || iy >= V.h[0]) return false;
            const float img_depth = full_depth(D, V.n_levels, ix, iy);
            if (img_depth >= kMaxDepth) { if (view_block) return false; continue; }
            const float diff = img_depth - depth;
            if (!view_block) {
                const double a = abs_int ? (double)abs((int)diff) : (double)fabsf(diff);
                if (!(a < thr)) return false;
            } else if (!((double)diff > thr)) return false;
        }
    return true;
}

}  // namespace

__global__ void __launch_bounds__(256) depth_fill_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = kMaxDepth;
}

// Scene::setDepths(patch, subtract = false): one thread per (patch, attached image)
__global__ void __launch_bounds__(256) set_depths_kernel(DevScene sc, const DevDepthView* __restrict__ depths, DevBatch b) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = (int)(t / b.max_images), k = (int)(t - (long long)p * b.max_images);
    if (p >= b.n) return;
    if (b.ok && !b.ok[p]) return;
    const int nimg = b.n_images[p];
    if (k >= nimg || k >= b.max_images) return;
    const int idx = b.images[(size_t)p * b.max_images + k];
    if (idx < 0 || idx >= sc.n_views) return;
    const DevView& V = sc.views[idx];
    const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
    const float scale = b.scale[p];
    const DevDepthView& D = depths[idx];
    int level, x, y;
    float d;
    if (!set_depths_cell(sc, V, D, c, scale, level, x, y, d)) return;
    // `if (d < old) old = d` for non-negative floats = unsigned minimum of the bit patterns
    atomicMin((unsigned int*)&D.d[level][y + (size_t)x * D.rows[level]], __float_as_uint(d));
}

// depthTests / viewBlockTest / pixelFreeTests: one wavefront per patch, one lane per view
__global__ void __launch_bounds__(256) depth_gates_kernel(DevScene sc, const DevDepthView* __restrict__ depths, DevBatch b,
                                                          float margin, int abs_int, int32_t* __restrict__ n_visible,
                                                          int32_t* __restrict__ n_blocking, int32_t* __restrict__ n_free) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < b.n; p += nwaves) {
        const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
        const float n[3] = {b.normal[4 * p], b.normal[4 * p + 1], b.normal[4 * p + 2]};
        const float scale = b.scale[p];
        int nimg = b.n_images[p];
        nimg = nimg < 0 ? 0 : (nimg > b.max_images ? b.max_images : nimg);
        // the attached images: depthTests (similar depth) and pixelFreeTests
        int nv = 0, nf = 0;
        for (int k0 = 0; k0 < nimg; k0 += 64) {   // (lists longer than a wavefront: HPMVS_MAX_IMAGES = 256)
        bool vis = false, fre = false;
        if (k0 + lane < nimg) {
            const int idx = b.images[(size_t)p * b.max_images + k0 + lane];
            if (idx >= 0 && idx < sc.n_views) {
                const DevView& V = sc.views[idx];
                vis = depth_test(V, depths[idx], c, n, scale, margin, false, abs_int);
                const int level = level_rounded(V, c, scale);
                if (level >= 0 && level < V.n_levels) {
                    float r[3];
                    project_level(V, c, level, r);
                    const int ix = cvt_i32_x86((double)(r[0] / r[2]) + 0.5), iy = cvt_i32_x86((double)(r[1] / r[2]) + 0.5);
                    if (!(ix < 0 || ix >= V.w[level] || iy < 0 || iy >= V.h[level])) {
                        const DevDepthView& D = depths[idx];
                        int fl, x, y;
                        float dv = kMaxDepth;   // (a pixel inside the image whose cell lies outside the map: getDetphAtLevel returns MAX_DEPTH)
                        if (pixel_free_cell(V, D, c, scale, fl, x, y)) dv = D.d[fl][y + (size_t)x * D.rows[fl]];
                        fre = dv == kMaxDepth;
                    }
                }
            }
        }
        nv += __popcll(__ballot(vis)); nf += __popcll(__ballot(fre));
        }
        // every view of the scene: does the patch lie in front of a confirmed surface there?
        int nb = 0;
        for (int base = 0; base < sc.n_views; base += 64) {
            const int v = base + lane;
            bool blk = false;
            if (v < sc.n_views) {
                const DevView& V = sc.views[v];
                if (view_block_visible(V, c, scale)) blk = depth_test(V, depths[v], c, n, scale, margin, true, abs_int);
            }
            nb += __popcll(__ballot(blk));
        }
        if (lane == 0) {
            if (n_visible) n_visible[p] = nv;
            if (n_blocking) n_blocking[p] = nb;
            if (n_free) n_free[p] = nf;
        }
    }
}

// The depth-map cells a candidate's gates READ and the cells setDepths would WRITE for it, as integers -- what a scheduler
// needs to run the candidates of one priority level in conflict-free waves and still get the reference's sequential
// result (hpmvs_amd/frontier.py; the gates of candidate i must see the writes of the candidates accepted before it,
// CellProcessor.cpp:130-142 -> Scene.cpp:351-381).  Computed by the very helpers the gates / setDepths kernels use.
//   wr [n][M][4]  attached image k: view, level, x, y of the cell setDepths offers its depth to   (view = -1: none)
//   fr [n][M][4]  attached image k: view, level, x, y of the cell pixelFreeTest reads             (view = -1: none)
//   at [n][M][3]  attached image k: view, ix0, iy0: top-left of the 3x3 level-0 pixel block depthTests reads through
//                 getFullDepth (every pyramid level at (pixel / 2) >> level)                      (view = -1: none)
//   vb [n][V][3]  every view v of the scene: 1 if viewBlockTest looks at it, then ix0, iy0 of its 3x3 block
__global__ void __launch_bounds__(256) depth_footprints_kernel(DevScene sc, const DevDepthView* __restrict__ depths, DevBatch b,
                                                               int32_t* __restrict__ wr, int32_t* __restrict__ fr,
                                                               int32_t* __restrict__ at, int32_t* __restrict__ vb) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int M = b.max_images;
    for (int p = wave; p < b.n; p += nwaves) {
        const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
        const float scale = b.scale[p];
        int nimg = b.n_images[p];
        nimg = nimg < 0 ? 0 : (nimg > M ? M : nimg);
        for (int kk = lane; kk < M; kk += 64) {
            int w4[4] = {-1, 0, 0, 0}, f4[4] = {-1, 0, 0, 0}, a3[3] = {-1, 0, 0};
            const int idx = kk < nimg ? b.images[(size_t)p * M + kk] : -1;
            if (idx >= 0 && idx < sc.n_views) {
                const DevView& V = sc.views[idx];
                const DevDepthView& D = depths[idx];
                int level, x, y;
                float d;
                if (set_depths_cell(sc, V, D, c, scale, level, x, y, d)) { w4[0] = idx; w4[1] = level; w4[2] = x; w4[3] = y; }
                if (pixel_free_cell(V, D, c, scale, level, x, y)) { f4[0] = idx; f4[1] = level; f4[2] = x; f4[3] = y; }
                float depth;
                depth_test_block(V, c, a3[1], a3[2], depth);
                a3[0] = idx;
            }
            const size_t o = (size_t)p * M + kk;
            if (wr) for (int k = 0; k < 4; k++) wr[4 * o + k] = w4[k];
            if (fr) for (int k = 0; k < 4; k++) fr[4 * o + k] = f4[k];
            if (at) for (int k = 0; k < 3; k++) at[3 * o + k] = a3[k];
        }
        if (vb)
            for (int base = 0; base < sc.n_views; base += 64) {
                const int v = base + lane;
                if (v < sc.n_views) {
                    const DevView& V = sc.views[v];
                    int ix0 = 0, iy0 = 0;
                    float depth;
                    const int seen = view_block_visible(V, c, scale) ? 1 : 0;
                    if (seen) depth_test_block(V, c, ix0, iy0, depth);
                    const size_t o = ((size_t)p * sc.n_views + v) * 3;
                    vb[o] = seen; vb[o + 1] = ix0; vb[o + 2] = iy0;
                }
            }
    }
}

// ---- Scene::setDepths(patch, subtract) for a whole batch with the result of the SEQUENTIAL loop (round 6).
// subtract = false is a minimum per cell -- any order gives the same maps (set_depths_kernel) -- but subtract = true
// (Scene.cpp:373-376: a cell that holds exactly this patch's depth becomes MAX_DEPTH again; CellProcessor::branch takes the
// depths of the patch of a leaf it splits back, CellProcessor.cpp:276-279, and enters the children's) is not: whether a cell
// ends at MAX_DEPTH, at the old patch's depth or at a child's depends on the order of the calls that touch it.  So a batch
// with subtractions is applied cell by cell in call order: every (patch, attached image) that reaches a cell becomes a key
// (cell index in the scene's map pool << 32 | call index), the keys are sorted, and the thread that finds the first key of a
// cell replays that cell's calls in order.  Cells are independent, calls on one cell are sequential: the sequential loop's maps.
__global__ void __launch_bounds__(256) depth_ops_keys_kernel(DevScene sc, const DevDepthView* __restrict__ depths, const float* pool, DevBatch b,
                                                             unsigned long long* __restrict__ keys, unsigned int* __restrict__ counter) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = (int)(t / b.max_images), k = (int)(t - (long long)p * b.max_images);
    bool valid = false;
    unsigned long long key = 0ull;
    if (p < b.n && !(b.ok && !b.ok[p])) {
        const int nimg = b.n_images[p];
        if (k < nimg && k < b.max_images) {
            const int idx = b.images[(size_t)p * b.max_images + k];
            if (idx >= 0 && idx < sc.n_views) {
                const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
                const DevDepthView& D = depths[idx];
                int level, x, y;
                float d;
                if (set_depths_cell(sc, sc.views[idx], D, c, b.scale[p], level, x, y, d)) {
                    const size_t cell = (size_t)(&D.d[level][y + (size_t)x * D.rows[level]] - pool);
                    key = ((unsigned long long)cell << 32) | (unsigned long long)(unsigned int)t;
                    valid = true;
                }
            }
        }
    }
    // one atomic per wavefront; the order of the keys in the buffer does not matter (they are sorted next)
    const unsigned long long m = __ballot(valid);
    if (m) {
        const int lane = lane_id();
        unsigned int base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (unsigned int)__popcll(m));
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
        if (valid) keys[base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = key;
    }
}
__global__ void __launch_bounds__(256) depth_ops_apply_kernel(DevScene sc, const DevDepthView* __restrict__ depths, float* pool, DevBatch b,
                                                              const uint8_t* __restrict__ subtract, const unsigned long long* __restrict__ keys,
                                                              unsigned int count) {
    const unsigned int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const unsigned long long cell = keys[j] >> 32;
    if (j > 0 && (keys[j - 1] >> 32) == cell) return;   // not the first call on this cell
    float v = pool[cell];
    for (unsigned int q = j; q < count; q++) {
        const unsigned long long key = keys[q];
        if ((key >> 32) != cell) break;
        const unsigned int t = (unsigned int)key;
        const int p = (int)(t / (unsigned int)b.max_images), k = (int)(t - (unsigned int)p * (unsigned int)b.max_images);
        const int idx = b.images[(size_t)p * b.max_images + k];
        const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
        int level, x, y;
        float d;
        set_depths_cell(sc, sc.views[idx], depths[idx], c, b.scale[p], level, x, y, d);   // (the same cell; its depth)
        if (subtract && subtract[p]) { if (v == d) v = kMaxDepth; }   // Scene.cpp:373-374
        else if (d < v) v = d;                                        // :375-376
    }
    pool[cell] = v;
}
// Scene::getLevelSupport (Scene.cpp:334-343): the attached images with std::round(getLevel) > minLevel, i.e. (float)log2(r) >=
// minLevel + 0.5 -- the threshold DevScene::lvl_thr[minLevel] was bisected for against the host's log2 (photometric.hpp)
__global__ void __launch_bounds__(256) level_support_kernel(DevScene sc, DevBatch b, int min_level, int32_t* __restrict__ support) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n) return;
    const float c[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
    const float scale = b.scale[p];
    int nimg = b.n_images[p];
    nimg = nimg < 0 ? 0 : (nimg > b.max_images ? b.max_images : nimg);
    int n = 0;
    for (int k = 0; k < nimg; k++) {
        const int idx = b.images[(size_t)p * b.max_images + k];
        if (idx < 0 || idx >= sc.n_views) continue;
        const DevView& V = sc.views[idx];
        if (min_level >= 0 && min_level < kMaxLevels - 1) {
            const float d0 = c[0] - V.center[0], d1 = c[1] - V.center[1], d2 = c[2] - V.center[2];
            const float fz = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
            const double r = (double)(scale * V.fsum) / (2.0 * (double)fz);
            n += (r >= sc.lvl_thr[min_level]) ? 1 : 0;
        } else n += (level_rounded(V, c, scale) > min_level) ? 1 : 0;
    }
    support[p] = n;
}

void launch_level_support(const DevScene& sc, const DevBatch& b, int min_level, int32_t* support, hipStream_t st) {
    if (b.n <= 0) return;
    hipLaunchKernelGGL(level_support_kernel, dim3((unsigned)((b.n + 255) / 256)), dim3(256), 0, st, sc, b, min_level, support);
}
void launch_depth_ops_keys(const DevScene& sc, const DevDepthView* depths, const float* pool, const DevBatch& b, unsigned long long* keys,
                           unsigned int* counter, hipStream_t st) {
    if (b.n <= 0) return;
    const size_t threads = (size_t)b.n * (size_t)b.max_images;
    hipLaunchKernelGGL(depth_ops_keys_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, sc, depths, pool, b, keys, counter);
}
// sorts `count` keys from keys_in to keys_out; temp == nullptr: only reports the temporary bytes rocPRIM needs
int depth_ops_sort(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, unsigned int count, hipStream_t st) {
    return (int)rocprim::radix_sort_keys(temp, *temp_bytes, keys_in, keys_out, (size_t)count, 0u, 64u, st);
}
void launch_depth_ops_apply(const DevScene& sc, const DevDepthView* depths, float* pool, const DevBatch& b, const uint8_t* subtract,
                            const unsigned long long* keys, unsigned int count, hipStream_t st) {
    if (!count) return;
    hipLaunchKernelGGL(depth_ops_apply_kernel, dim3((count + 255) / 256), dim3(256), 0, st, sc, depths, pool, b, subtract, keys, count);
}

void launch_depth_footprints(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, int32_t* wr, int32_t* fr, int32_t* at,
                             int32_t* vb, hipStream_t st) {
    if (b.n <= 0) return;
    int blocks = (b.n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(depth_footprints_kernel, dim3(blocks), dim3(256), 0, st, sc, depths, b, wr, fr, at, vb);
}

void launch_depth_fill(float* p, size_t n, hipStream_t st) {
    if (!n) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(depth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
}
void launch_set_depths(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, hipStream_t st) {
    if (b.n <= 0) return;
    const size_t threads = (size_t)b.n * (size_t)b.max_images;
    hipLaunchKernelGGL(set_depths_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, sc, depths, b);
}
void launch_depth_gates(const DevScene& sc, const DevDepthView* depths, const DevBatch& b, float margin, int abs_int,
                        int32_t* n_visible, int32_t* n_blocking, int32_t* n_free, hipStream_t st) {
    if (b.n <= 0) return;
    int blocks = (b.n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(depth_gates_kernel, dim3(blocks), dim3(256), 0, st, sc, depths, b, margin, abs_int, n_visible,
                       n_blocking, n_free);
}

}  // namespace hpmvs
