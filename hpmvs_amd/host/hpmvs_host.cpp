// hpmvs_host.cpp -- C++ mirror of the reference's public interface for the refinement path
// (mo3d::Patch3d, Camera, Image, Scene view, PatchOptimizer) on top of the C ABI
// (include/hpmvs_amd.h).  Host logic only: scene assembly and marshalling; every piece of the hot
// path's arithmetic runs in the gfx950 kernels behind hpmvs_optimize_batch.
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <strings.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <omp.h>

#include <hpmvs/Camera.h>
#include <hpmvs/HpmvsOptions.h>
#include <hpmvs/Image.h>
#include <hpmvs/Patch3d.h>
#include <hpmvs/PatchOptimizer.h>
#include <hpmvs/PlyWriter.h>
#include <hpmvs/Scene.h>

#include "../../include/hpmvs_amd.h"

// ---------------------------------------------------------------- where the structure-of-arrays copies of a batch live
// The C ABI uses host arrays that lie in pinned, mapped host memory IN PLACE (include/hpmvs_amd.h, "Host-pointer calls and
// pinned memory"): no device allocation, staging copy or free per array and call.  PVec<T> is a std::vector whose storage
// comes from hpmvs_host_alloc through a per-thread cache of power-of-two blocks (pinning pages is slow: a block is pinned
// once and then reused by later calls of the thread; it goes back to the system when the thread ends).  Batches up to
// kPinFrom patches take ordinary memory: the C ABI serves those from its own pinned staging pool anyway.
namespace {
constexpr size_t kPinFrom = 4096;
// A thread keeps at most this many bytes of pinned blocks for reuse ($HPMVS_PIN_CACHE_MB, default 512): a frontier loop whose
// round size changes every round would otherwise leave one non-swappable block per size class and array behind (images[]
// alone is m * 64 * 4 bytes) and end up holding several times its peak working set pinned.  Beyond the cap the
// least recently returned blocks go back to the system (hipHostFree).
struct PinnedCache {
    struct Entry { void* p; int cls; size_t bytes; };
    std::vector<Entry> lru_;  // blocks available for reuse, oldest first
    size_t cached_ = 0, cap_;
    PinnedCache() {
        const char* e = getenv("HPMVS_PIN_CACHE_MB");
        const long mb = e ? atol(e) : 512;
        cap_ = (size_t)(mb < 0 ? 0 : mb) << 20;
    }
    ~PinnedCache() { for (auto& e : lru_) hpmvs_host_free(e.p); }
    // size classes: four per octave from 4 KB (a block is at most 25 % larger than asked for)
    static int cls(size_t bytes, size_t* rounded) {
        int e = 12;
        while (((size_t)1 << (e + 1)) < bytes) e++;
        if (bytes <= ((size_t)1 << 12)) { *rounded = (size_t)1 << 12; return 0; }
        const size_t q = (size_t)1 << (e - 2);  // a quarter of 2^e: classes 2^e + k * q, k = 1..4
        const size_t k = (bytes - ((size_t)1 << e) + q - 1) / q;
        *rounded = ((size_t)1 << e) + k * q;
        return 4 * (e - 12) + (int)k;
    }
    void* get(size_t bytes, size_t* rounded) {
        const int c = cls(bytes, rounded);
        for (size_t i = lru_.size(); i-- > 0;)  // the most recently returned block of the class
            if (lru_[i].cls == c) { void* p = lru_[i].p; cached_ -= lru_[i].bytes; lru_.erase(lru_.begin() + (long)i); return p; }
        return hpmvs_host_alloc(*rounded);
    }
    // Over the cap the least recently returned blocks go back to the system -- but never the block that was returned last (ADVICE
    // r5): a working set whose single blocks are larger than the cap (a 1e7-patch batch: images[] alone is GBs) would otherwise be
    // unpinned and re-pinned by every call, hundreds of milliseconds each; this way the cap follows the last call's largest block.
    // `keep_newest` entries at the back survive; the survivors are compacted once, not erased one by one from the front.
    void trim(size_t keep_newest) {
        size_t drop = 0;
        while (cached_ > cap_ && drop + keep_newest < lru_.size()) { cached_ -= lru_[drop].bytes; hpmvs_host_free(lru_[drop].p); drop++; }
        if (drop) lru_.erase(lru_.begin(), lru_.begin() + (long)drop);
    }
    void put(void* p, size_t rounded) {
        size_t r;
        lru_.push_back(Entry{p, cls(rounded, &r), rounded});
        cached_ += rounded;
        trim(1);
    }
    size_t cached_bytes() const { return cached_; }
    void set_cap(size_t bytes) { cap_ = bytes; trim(0); }
};
thread_local PinnedCache tl_pins;
// Whether a block is pinned is recorded IN the block (a 64-byte header in front of the data), not in the allocator object:
// a vector that owns a pinned block and later reallocates from an allocator whose pinning failed must still hand the old
// block back to the cache, not to ::operator delete.
struct PinHeader { unsigned long long magic; unsigned long long pinned; size_t rounded; char pad[40]; };
static_assert(sizeof(PinHeader) == 64, "the data keeps the block's 64-byte alignment");
constexpr unsigned long long kPinMagic = 0x48504d5650494e21ull;  // "HPMVPIN!"
template <class T>
struct PinnedAlloc {
    using value_type = T;
    bool pin;  // false: ordinary memory is asked for
    explicit PinnedAlloc(bool p = false) : pin(p) {}
    template <class U> PinnedAlloc(const PinnedAlloc<U>& o) : pin(o.pin) {}
    T* allocate(size_t n) {
        const size_t bytes = sizeof(T) * n + sizeof(PinHeader);
        PinHeader* h = nullptr;
        size_t rounded = 0;
        if (pin) h = (PinHeader*)tl_pins.get(bytes, &rounded);  // (nullptr: pinning failed -> pageable memory, staged by the C ABI)
        const bool pinned = h != nullptr;
        if (!h) h = (PinHeader*)::operator new(bytes);
        h->magic = kPinMagic; h->pinned = pinned ? 1ull : 0ull; h->rounded = rounded;
        return (T*)(h + 1);
    }
    void deallocate(T* p, size_t) {
        PinHeader* h = (PinHeader*)p - 1;
        if (h->magic != kPinMagic) { std::fprintf(stderr, "hpmvs_host: PinnedAlloc::deallocate of a foreign block\n"); std::abort(); }
        if (h->pinned) tl_pins.put(h, h->rounded); else ::operator delete(h);
    }
    template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }   // any instance can free any block
    template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T> using PVec = std::vector<T, PinnedAlloc<T>>;
}  // namespace

namespace mo3d {

// ---------------------------------------------------------------- Patch3d (reference src/hpmvs/Patch3d.cpp:28-74)
std::atomic_uint_least32_t Patch3d::runningIdx(0);

Patch3d::Patch3d() {
    uid_ = generateUid();
    center_ = Eigen::Vector4f::Zero();
    normal_ = Eigen::Vector4f::Zero();
    color_ = Eigen::Vector3f::Zero();
    images_.clear();
    scale_3dx_ = 0.0;
    dscale_ = 0.0;
    ncc_ = 0.0;
    priorityReduction_ = 0;
    expanded_ = false;
    flatness_ = 0.0;
    dirty_ = false;
}
Patch3d::Patch3d(const Patch3d& other) { uid_ = generateUid(); *this = other; }
Patch3d::~Patch3d() {}
Patch3d& Patch3d::operator=(const Patch3d& s) {
    center_ = s.center_; normal_ = s.normal_; color_ = s.color_; images_ = s.images_;
    scale_3dx_ = s.scale_3dx_; dscale_ = s.dscale_; ncc_ = s.ncc_;
    priorityReduction_ = s.priorityReduction_; expanded_ = s.expanded_; flatness_ = s.flatness_; dirty_ = s.dirty_;
    return *this;
}

// ---------------------------------------------------------------- Camera (reference src/hpmvs/Camera.cpp:34-99)
void Camera::init(const mo3d::NVM_Camera* cam, int width, int height, const int maxLevel) {
    hpmvs_camera hc;
    const double q[4] = {cam->rq[0], cam->rq[1], cam->rq[2], cam->rq[3]};
    const double c[3] = {cam->c[0], cam->c[1], cam->c[2]};
    if (hpmvs_camera_from_nvm(cam->f, q, c, width, height, maxLevel, &hc) != HPMVS_OK) {
        std::cerr << "Camera::init: " << hpmvs_last_error() << std::endl;
        exit(1);
    }
    name_ = cam->filename;
    projection_.resize(maxLevel + 1);
    kMat_.resize(maxLevel + 1);
    memcpy(projection_[0].m, hc.P0, sizeof(float) * 12);
    const float K0[9] = {(float)cam->f, 0, (float)(width / 2.0), 0, (float)cam->f, (float)(height / 2.0), 0, 0, 1};
    memcpy(kMat_[0].m, K0, sizeof(K0));
    for (int l = 1; l < maxLevel + 1; l++) {  // rows 0,1 halved per level (Camera.cpp:55-63)
        projection_[l] = projection_[l - 1];
        kMat_[l] = kMat_[l - 1];
        for (int k = 0; k < 8; k++) projection_[l].m[k] /= 2.0;
        for (int k = 0; k < 6; k++) kMat_[l].m[k] /= 2.0;
    }
    center_ = Eigen::Vector4f(hc.center[0], hc.center[1], hc.center[2], hc.center[3]);
    const float* r2 = &hc.P0[8];
    const float nrm = std::sqrt((r2[0] * r2[0] + r2[1] * r2[1]) + r2[2] * r2[2]);
    oAxis_ = Eigen::Vector4f(r2[0] / nrm, r2[1] / nrm, r2[2] / nrm, r2[3] / nrm);
    xAxis_ = Eigen::Vector3f(hc.xaxis[0], hc.xaxis[1], hc.xaxis[2]);
    yAxis_ = Eigen::Vector3f(hc.yaxis[0], hc.yaxis[1], hc.yaxis[2]);
    zAxis_ = Eigen::Vector3f(hc.zaxis[0], hc.zaxis[1], hc.zaxis[2]);
    const float* r0 = &hc.P0[0];
    const float* r1 = &hc.P0[4];
    ipscale_ = (std::sqrt((r0[0] * r0[0] + r0[1] * r0[1]) + r0[2] * r0[2]) +
                std::sqrt((r1[0] * r1[0] + r1[1] * r1[1]) + r1[2] * r1[2])) / 2.0;
}

Eigen::Vector3f Camera::project(const Eigen::Vector4f& X, const int level) const {  // Camera.h:45-62
    if (level >= (int)projection_.size()) { std::cerr << "Illegal level access in Camera"; exit(1); }
    const float* P = projection_[level].m;
    float r[3];
    for (int k = 0; k < 3; k++) r[k] = ((P[k * 4] * X[0] + P[k * 4 + 1] * X[1]) + P[k * 4 + 2] * X[2]) + P[k * 4 + 3] * X[3];
    if (r[2] <= 0.0f) return Eigen::Vector3f(-0xffff, -0xffff, -1.0f);
    const float lo = (float)(INT_MIN + 3.0f), hi = (float)(INT_MAX - 3.0f);
    return Eigen::Vector3f(std::max(lo, std::min(hi, r[0] / r[2])), std::max(lo, std::min(hi, r[1] / r[2])), 1.0f);
}
float Camera::getScale(const Eigen::Vector4f& coord, const int level) const {
    const float fz = (coord - center_).norm();
    const float ftmp = kMat_[0](0, 0) + kMat_[0](1, 1);
    if (ftmp == 0.0) return 1.0;
    return 2.0 * fz * (0x0001 << level) / ftmp;
}
float Camera::getLevel(const Eigen::Vector4f& coord, const float scale) const {
    const float fz = (coord - center_).norm();
    return std::log2(scale * (float)(kMat_[0](0, 0) + kMat_[0](1, 1)) / (2.0 * fz));
}
int Camera::getLeveli(const Eigen::Vector4f& coord, const float scale, const int maxLevel) const {
    return std::max(0, std::min(maxLevel, (int)std::round(getLevel(coord, scale))));
}

// ---------------------------------------------------------------- Image (reference src/hpmvs/Image.cpp:33-66)
void Image::init(const mo3d::NVM_Camera* cam, const int maxLevel) {
    path_ = cam->filename;
    f_ = cam->f;
    k1_ = cam->r;
    maxLevel_ = std::max(1, maxLevel);
}
void Image::setPixels(int width, int height, const uint8_t* rgb) {
    width_ = width; height_ = height;
    rgb_.assign(rgb, rgb + (size_t)width * height * 3);
}
bool Image::load() {
    if (!rgb_.empty()) return true;  // pixels were handed over with setPixels
    if (k1_ != 0) { std::cerr << "Image::load: radial undistortion (k1 != 0) is outside this path\n"; return false; }
    std::ifstream in(path_.c_str(), std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    in >> magic >> w >> h >> maxv;
    if (!in || magic != "P6" || maxv != 255 || w <= 0 || h <= 0) return false;
    in.get();
    rgb_.resize((size_t)w * h * 3);
    in.read((char*)rgb_.data(), rgb_.size());
    if (!in) return false;
    width_ = w; height_ = h;
    return true;
}

// ---------------------------------------------------------------- NVM_V3 I/O (reference src/hpmvs/NVMReader.cpp:31-182)
static bool read_model(std::istream& in, NVM_Model& m) {
    int nc = 0, np = 0;
    if (!(in >> nc)) return false;
    m.cameras.resize(nc > 0 ? nc : 0);
    for (int i = 0; i < nc; i++) {
        NVM_Camera& c = m.cameras[i];
        int check = 0;
        in >> c.filename >> c.f >> c.rq[0] >> c.rq[1] >> c.rq[2] >> c.rq[3] >> c.c[0] >> c.c[1] >> c.c[2] >> c.r >> check;
        for (char& ch : c.filename) if (ch == '"') ch = ' ';
    }
    if (nc > 0) in >> np;
    m.points.resize(np > 0 ? np : 0);
    for (int i = 0; i < np; i++) {
        NVM_Point& p = m.points[i];
        int nm = 0;
        in >> p.xyz[0] >> p.xyz[1] >> p.xyz[2] >> p.rgb[0] >> p.rgb[1] >> p.rgb[2] >> nm;
        p.measurements.resize(nm > 0 ? nm : 0);
        for (int k = 0; k < nm; k++)
            in >> p.measurements[k].imgIndex >> p.measurements[k].featIndex >> p.measurements[k].xy[0] >> p.measurements[k].xy[1];
    }
    return true;
}

void NVMReader::readFile(const char* path, std::vector<NVM_Model>& models, bool fixPath) {
    models.clear();
    std::ifstream in(path);
    if (!in.good()) { std::cerr << "cannot read from <" << path << ">" << std::endl; return; }
    std::string folder(path);
    const size_t slash = folder.find_last_of('/');
    folder = (slash == std::string::npos) ? std::string() : folder.substr(0, slash + 1);
    std::string header;
    in >> header;
    if (strcasecmp("NVM_V3", header.c_str()) != 0) { std::cerr << "<" << path << "> is no valid nvm file" << std::endl; return; }
    do {
        models.emplace_back();
        if (!read_model(in, models.back())) break;
        if (fixPath)
            for (NVM_Camera& c : models.back().cameras)
                if (!c.filename.empty() && c.filename[0] != '/') c.filename = folder + c.filename;
    } while (in.good() && models.back().cameras.size() > 0);
    if (!models.empty()) models.pop_back();  // the terminating empty model
}

void NVMReader::saveNVM(const char* path, const std::vector<NVM_Model>& models) {
    std::ofstream out(path);
    if (!out.good()) { std::cerr << "cannot write to <" << path << ">" << std::endl; return; }
    out << std::setprecision(12);
    out << "NVM_V3" << std::endl;
    for (const NVM_Model& m : models) {
        out << std::endl << m.cameras.size() << std::endl;
        for (const NVM_Camera& c : m.cameras)
            out << c.filename << " " << c.f << " " << c.rq[0] << " " << c.rq[1] << " " << c.rq[2] << " " << c.rq[3] << " "
                << c.c[0] << " " << c.c[1] << " " << c.c[2] << " " << c.r << " " << 0 << std::endl;
        if (!m.cameras.empty()) out << std::endl << m.points.size() << std::endl;
        for (const NVM_Point& p : m.points) {
            out << p.xyz[0] << " " << p.xyz[1] << " " << p.xyz[2] << " " << (int)p.rgb[0] << " " << (int)p.rgb[1] << " "
                << (int)p.rgb[2] << " " << p.measurements.size();
            for (const NVM_Measurement& q : p.measurements)
                out << " " << q.imgIndex << " " << q.featIndex << " " << q.xy[0] << " " << q.xy[1];
            out << std::endl;
        }
    }
    out << "0";
}

// ---------------------------------------------------------------- extended PLY (reference include/hpmvs/doctree.h:526-622)
bool writeExtPly(const char* name, const std::vector<Ppatch3d>& patches, bool binary, bool normal, bool scale,
                 bool visibility) {
    {
        std::ofstream h(name, std::ofstream::out);
        if (!h.good()) return false;
        h << "ply" << std::endl;
        h << (binary ? "format binary_little_endian 1.0" : "format ascii 1.0") << std::endl;
        h << "element vertex " << (int)patches.size() << std::endl;
        h << "property float x" << std::endl << "property float y" << std::endl << "property float z" << std::endl;
        if (normal) h << "property float nx" << std::endl << "property float ny" << std::endl << "property float nz" << std::endl;
        h << "property uchar red" << std::endl << "property uchar green" << std::endl << "property uchar blue" << std::endl;
        if (scale) h << "property float scalar_scale" << std::endl;
        if (visibility) {
            h << "element point_visibility " << (int)patches.size() << std::endl;
            h << "property list uint uint visible_cameras" << std::endl;
        }
        h << "end_header" << std::endl;
    }
    std::ofstream d(name, binary ? std::ofstream::binary | std::ofstream::app : std::ofstream::app);
    for (const Ppatch3d& p : patches) {
        const unsigned char col[3] = {(unsigned char)p->color_[0], (unsigned char)p->color_[1], (unsigned char)p->color_[2]};
        if (binary) {
            const float v[3] = {p->x(), p->y(), p->z()};
            d.write((const char*)v, sizeof(v));
            if (normal) { const float nn[3] = {p->normal_[0], p->normal_[1], p->normal_[2]}; d.write((const char*)nn, sizeof(nn)); }
            d.write((const char*)col, 3);
            if (scale) d.write((const char*)&p->scale_3dx_, sizeof(float));
        } else {
            d << p->x() << " " << p->y() << " " << p->z() << " ";
            if (normal) d << p->normal_[0] << " " << p->normal_[1] << " " << p->normal_[2] << " ";
            d << (int)col[0] << " " << (int)col[1] << " " << (int)col[2] << " ";
            if (scale) d << p->scale_3dx_ << " ";
            d << std::endl;
        }
    }
    if (visibility)
        for (const Ppatch3d& p : patches) {
            if (binary) {
                const uint32_t n = (uint32_t)p->images_.size();
                d.write((const char*)&n, 4);
                for (int id : p->images_) { const uint32_t u = (uint32_t)id; d.write((const char*)&u, 4); }
            } else {
                d << (int)p->images_.size() << " ";
                for (int id : p->images_) d << (uint32_t)id << " ";
                d << std::endl;
            }
        }
    return d.good();
}

// ---------------------------------------------------------------- Scene view
Scene::Scene() : dev_(nullptr), device_(0), maxLevel_(5) {
    if (const char* e = getenv("HPMVS_DEVICE")) device_ = atoi(e);
}
// ---------------------------------------------------------------- combiner of single-patch calls
// The reference's callers hand over ONE patch per optimize() call from many host threads.  A lone patch is a
// serial chain of ~90 (up to 1000) evaluation rounds of ~22 us on the GPU, so its latency cannot be hidden by
// the device; what can be done is to run the patches of all threads that are waiting AT THE SAME TIME in one
// launch instead of one launch (and one staging round trip) each.  Flat combining: a caller appends its patch
// to the pending list; the first caller that finds no collector becomes the collector, waits a moment for the
// other threads that were seen calling recently, takes the whole list, runs it as one optimizeBatch() and
// wakes the owners.  While it runs, the next arrivals elect the next collector, so several combined launches
// are in flight at once (each on its own staging entry / stream of the C ABI).  Results are those of
// optimizeBatch(): patches are independent, so combining changes no bit of any result.
namespace {
struct Combiner {
    struct Entry { Patch3d* patch; const HpmvsOptions* opt; bool done = false; uint8_t ok = 0; double f = 0.0; int evals = 0; };
    std::mutex mu;
    std::condition_variable cv_collect, cv_done;
    std::vector<Entry*> pending;
    bool collecting = false;
    int inside = 0;      // threads currently inside optimize()
    int launches = 0;    // combined batches currently on the device
    double peak = 1.0;   // decayed maximum of `inside`: how many threads call concurrently
};
// at most this many combined launches in flight: the device runs only a few kernels of different streams at once,
// so further callers are better off joining the next batch than queueing a launch of their own
#ifndef HPMVS_MAX_COMBINED_LAUNCHES
#define HPMVS_MAX_COMBINED_LAUNCHES 2
#endif
// (measured with 6 in flight, round 3: 256 threads 20.5 k -> 14.6 k patches/s, 512 threads 34.0 k -> 24.3 k: the
//  streams share a handful of hardware queues, so a further launch often waits for a whole earlier one)
constexpr int kMaxCombinedLaunches = HPMVS_MAX_COMBINED_LAUNCHES;

// two option sets that drive the device identically (the reference's callers hold one HpmvsOptions copy per thread)
bool same_options(const HpmvsOptions& a, const HpmvsOptions& b) {
    return a.MAXLEVEL == b.MAXLEVEL && a.MINLEVEL == b.MINLEVEL && a.MAX_ANGLE == b.MAX_ANGLE && a.MIN_ANGLE == b.MIN_ANGLE &&
           a.MAX_IMAGES_PER_PATCH == b.MAX_IMAGES_PER_PATCH && a.MIN_IMAGES_PER_PATCH == b.MIN_IMAGES_PER_PATCH &&
           a.NCC_ALPHA_1 == b.NCC_ALPHA_1 && a.NCC_ALPHA_2 == b.NCC_ALPHA_2;
}
}  // namespace

void* Scene::combiner() const {
    std::lock_guard<std::mutex> lk(mu_);
    if (!combiner_) combiner_ = new Combiner();
    return combiner_;
}

Scene::~Scene() {
    if (dev_) hpmvs_scene_destroy(dev_);
    delete (Combiner*)combiner_;
}

bool Scene::addCameras(const NVM_Model& model, const HpmvsOptions& options) {
    maxLevel_ = options.MAXLEVEL;
    for (const NVM_Camera& cam : model.cameras) {
        const int id = (int)cameras_.size();
        cameras_.emplace_back();
        images_.emplace_back();
        dict_[cam.filename] = id;
    }
    for (size_t ii = 0; ii < model.cameras.size(); ii++) {
        const int camId = dict_[model.cameras[ii].filename];
        images_[camId].init(&model.cameras[ii], options.MAXLEVEL);
        if (!images_[camId].load()) return false;
        cameras_[camId].init(&model.cameras[ii], images_[camId].getWidth(), images_[camId].getHeight(), options.MAXLEVEL);
    }
    return true;
}

bool Scene::extractCoVisiblilty(const NVM_Model& model, const HpmvsOptions&) {
    const int nCams = (int)model.cameras.size();
    if (nCams != (int)cameras_.size()) return false;
    std::vector<int> vis((size_t)nCams * nCams, 0);
    for (const NVM_Point& p : model.points) {
        const int m = (int)p.measurements.size();
        // the reference counts by measurement POSITION, not camera id (Scene.cpp:260-264); kept
        for (int ii = 0; ii < m; ii++)
            for (int jj = 0; jj < m; jj++)
                if (ii != jj && ii < nCams && jj < nCams) vis[(size_t)ii * nCams + jj]++;
    }
    covis_.clear();
    covis_.resize(nCams);
    for (int ii = 0; ii < nCams; ii++)
        for (int jj = 0; jj < nCams; jj++)
            if (vis[(size_t)ii * nCams + jj] >= 50) covis_[ii].emplace_back(jj);
    return true;
}

// Row width of the image-id arrays handed to the C ABI.  The reference's lists are unbounded vectors; a list cannot hold more
// than the scene's views (addImages never attaches a view twice), so rows of min(views, HPMVS_MAX_IMAGES) ids hold every list
// that can come back -- but never fewer than HPMVS_FAST_IMAGES, and exactly that for up to four patches (the unchanged
// callers' optimize()): 64-id rows are what the open batch behind single-patch calls takes.
static thread_local bool tl_full_rows = false;
static int list_stride(size_t n_views, size_t n_patches, size_t longest_input) {
    size_t m = n_views > longest_input ? n_views : longest_input;
    if (n_patches <= 4 && longest_input <= (size_t)HPMVS_FAST_IMAGES) m = HPMVS_FAST_IMAGES;
    if (m < (size_t)HPMVS_FAST_IMAGES) m = HPMVS_FAST_IMAGES;
    if (m > (size_t)HPMVS_MAX_IMAGES) m = HPMVS_MAX_IMAGES;
    return (int)m;
}

bool Scene::initPatches(const NVM_Model& model, const HpmvsOptions& options, std::vector<Ppatch3d>& out) const {
    hpmvs_scene* dev = deviceScene();
    if (!dev) return false;
    const size_t n = model.points.size();
    if (n == 0) return true;
    const PinnedAlloc<char> pin(true);  // (this entry stages whatever the batch size: pinned arrays are used in place)
    PVec<double> xyz(3 * n, pin);
    PVec<int32_t> off(n + 1, 0, pin), img(pin);
    {
        size_t total = 0;
        for (size_t i = 0; i < n; i++) total += model.points[i].measurements.size();
        img.reserve(total ? total : 1);  // (one pinned block, not a series of growing ones)
    }
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) xyz[3 * i + k] = model.points[i].xyz[k];
        for (const NVM_Measurement& m : model.points[i].measurements) img.push_back(m.imgIndex);
        off[i + 1] = (int32_t)img.size();
    }
    size_t longest_in = 0;
    for (size_t i = 0; i < n; i++) longest_in = std::max(longest_in, (size_t)(off[i + 1] - off[i]));
    const int stride = list_stride(cameras_.size(), n, longest_in);
    PVec<float> center(4 * n, pin), normal(4 * n, pin), scale(n, pin), color(3 * n, pin), ncc(n, pin);
    PVec<int32_t> nimg(n, pin), images(n * stride, -1, pin);
    PVec<uint8_t> ok(n, pin);
    hpmvs_options o;
    o.MAXLEVEL = options.MAXLEVEL; o.MINLEVEL = options.MINLEVEL; o.MAX_ANGLE = options.MAX_ANGLE;
    o.MIN_ANGLE = options.MIN_ANGLE; o.MAX_IMAGES_PER_PATCH = options.MAX_IMAGES_PER_PATCH;
    o.MIN_IMAGES_PER_PATCH = options.MIN_IMAGES_PER_PATCH; o.NCC_ALPHA_1 = options.NCC_ALPHA_1;
    o.NCC_ALPHA_2 = options.NCC_ALPHA_2;
    hpmvs_patch_batch b;
    memset(&b, 0, sizeof(b));
    b.n = (int32_t)n; b.max_images = stride;
    b.center = center.data(); b.normal = normal.data(); b.scale = scale.data(); b.n_images = nimg.data();
    b.images = images.data(); b.ok = ok.data(); b.color = color.data(); b.ncc = ncc.data();
    if (hpmvs_init_patches_batch(dev, &o, options.START_LEVEL, (int)n, xyz.data(), off.data(),
                                 img.empty() ? nullptr : img.data(), &b, 0, nullptr) != HPMVS_OK) {
        std::cerr << "initPatches: " << hpmvs_last_error() << std::endl;
        return false;
    }
    for (size_t i = 0; i < n; i++) {
        if (!ok[i]) continue;
        Ppatch3d p(new Patch3d);
        for (int k = 0; k < 4; k++) { p->center_[k] = center[4 * i + k]; p->normal_[k] = normal[4 * i + k]; }
        p->scale_3dx_ = scale[i];
        p->images_.assign(images.begin() + i * stride, images.begin() + i * stride + nimg[i]);
        p->ncc_ = ncc[i];
        p->color_ = Eigen::Vector3f(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
        out.push_back(p);
    }
    return true;
}

hpmvs_scene* Scene::deviceScene() const {
    std::lock_guard<std::mutex> lk(mu_);
    if (dev_) return dev_;
    hpmvs_scene* s = nullptr;
    if (hpmvs_scene_create((int)cameras_.size(), device_, &s) != HPMVS_OK) {
        std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
        return nullptr;
    }
    for (size_t i = 0; i < cameras_.size(); i++) {
        const Camera& c = cameras_[i];
        hpmvs_camera hc;
        memcpy(hc.P0, c.projection_[0].m, sizeof(float) * 12);
        for (int k = 0; k < 4; k++) hc.center[k] = c.center_[k];
        for (int k = 0; k < 3; k++) { hc.xaxis[k] = c.xAxis_[k]; hc.yaxis[k] = c.yAxis_[k]; hc.zaxis[k] = c.zAxis_[k]; }
        hc.fsum = c.kMat_[0](0, 0) + c.kMat_[0](1, 1);
        hc.n_levels = c.getLevels();
        if (hpmvs_scene_set_view(s, (int)i, images_[i].getWidth(), images_[i].getHeight(), images_[i].pixels().data(), 0,
                                 &hc) != HPMVS_OK ||
            hpmvs_scene_set_covis(s, (int)i, covis_[i].data(), (int)covis_[i].size()) != HPMVS_OK) {
            std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
            hpmvs_scene_destroy(s);
            return nullptr;
        }
    }
    if (hpmvs_scene_commit(s) != HPMVS_OK) {
        std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
        hpmvs_scene_destroy(s);
        return nullptr;
    }
    dev_ = s;
    return dev_;
}

// ---------------------------------------------------------------- depth maps + acceptance gates (batched)
namespace {
struct HostBatch {  // SoA copy of a patch list for the C ABI (inputs only)
    PVec<float> center, normal, scale;
    PVec<int32_t> nimg, images;
    hpmvs_patch_batch b;
    static size_t longest(const Patch3d* const* patches, size_t n) {
        size_t m = 1;
        for (size_t i = 0; i < n; i++) m = patches[i]->images_.size() > m ? patches[i]->images_.size() : m;
        return m < (size_t)HPMVS_MAX_IMAGES ? m : (size_t)HPMVS_MAX_IMAGES;
    }
    HostBatch(const Patch3d* const* patches, size_t n)
        : center(4 * n, PinnedAlloc<float>(true)), normal(4 * n, PinnedAlloc<float>(true)), scale(n, PinnedAlloc<float>(true)),
          nimg(n, PinnedAlloc<int32_t>(true)), images(n * longest(patches, n), -1, PinnedAlloc<int32_t>(true)) {
        const size_t stride = longest(patches, n);   // (inputs only: the rows need not hold more than the longest list)
        for (size_t i = 0; i < n; i++) {
            const Patch3d& p = *patches[i];
            for (int k = 0; k < 4; k++) { center[4 * i + k] = p.center_[k]; normal[4 * i + k] = p.normal_[k]; }
            scale[i] = p.scale_3dx_;
            const size_t m = p.images_.size() < stride ? p.images_.size() : stride;
            nimg[i] = (int32_t)m;
            for (size_t k = 0; k < m; k++) images[i * stride + k] = p.images_[k];
        }
        memset(&b, 0, sizeof(b));
        b.n = (int32_t)n; b.max_images = (int32_t)stride;
        b.center = center.data(); b.normal = normal.data(); b.scale = scale.data(); b.n_images = nimg.data(); b.images = images.data();
    }
};
}  // namespace

bool Scene::resetDepths() const {
    hpmvs_scene* dev = deviceScene();
    if (!dev || hpmvs_scene_depth_reset(dev) != HPMVS_OK) { std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl; return false; }
    return true;
}
bool Scene::setDepths(const Patch3d* const* patches, size_t n) const {
    hpmvs_scene* dev = deviceScene();
    if (!dev) return false;
    if (n == 0) return true;
    HostBatch hb(patches, n);
    if (hpmvs_set_depths_batch(dev, &hb.b, 0, nullptr) != HPMVS_OK) { std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl; return false; }
    return true;
}
bool Scene::setDepths(const Patch3d* const* patches, size_t n, const uint8_t* subtract) const {
    if (!subtract) return setDepths(patches, n);
    hpmvs_scene* dev = deviceScene();
    if (!dev) return false;
    if (n == 0) return true;
    HostBatch hb(patches, n);
    PVec<uint8_t> sub(subtract, subtract + n, PinnedAlloc<uint8_t>(true));
    if (hpmvs_depth_ops_batch(dev, &hb.b, sub.data(), 0, nullptr) != HPMVS_OK) { std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl; return false; }
    return true;
}
bool Scene::levelSupport(const Patch3d* const* patches, size_t n, int minLevel, std::vector<int>& support) const {
    support.assign(n, 0);
    hpmvs_scene* dev = deviceScene();
    if (!dev) return false;
    if (n == 0) return true;
    HostBatch hb(patches, n);
    PVec<int32_t> out(n, PinnedAlloc<int32_t>(true));
    if (hpmvs_level_support_batch(dev, &hb.b, minLevel, out.data(), 0, nullptr) != HPMVS_OK) { std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl; return false; }
    for (size_t i = 0; i < n; i++) support[i] = out[i];
    return true;
}
bool Scene::depthGates(const Patch3d* const* patches, size_t n, float margin, std::vector<int>& nVisible,
                       std::vector<int>& nBlocking, std::vector<int>& nFree, bool absInt) const {
    nVisible.assign(n, 0); nBlocking.assign(n, 0); nFree.assign(n, 0);
    hpmvs_scene* dev = deviceScene();
    if (!dev) return false;
    if (n == 0) return true;
    HostBatch hb(patches, n);
    const PinnedAlloc<char> pin(true);
    PVec<int32_t> v(n, pin), k(n, pin), f(n, pin);
    if (hpmvs_depth_gates_batch(dev, &hb.b, margin, absInt ? 1 : 0, v.data(), k.data(), f.data(), 0, nullptr) != HPMVS_OK) {
        std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
        return false;
    }
    for (size_t i = 0; i < n; i++) { nVisible[i] = v[i]; nBlocking[i] = k[i]; nFree[i] = f[i]; }
    return true;
}

namespace {
// A depth-map cell as a key: (view, pyramid level, x, y)
inline uint64_t cell_key(int view, int level, int x, int y) {
    return ((uint64_t)(((unsigned)view << 3) | (unsigned)level) << 48) | ((uint64_t)((unsigned)x & 0xFFFFFFu) << 24) | (uint64_t)((unsigned)y & 0xFFFFFFu);
}
// What hpmvs_depth_footprints_batch returns, kept as it comes (PatchOptimizer::extendLevel walks these arrays directly: a level of
// 1e5 candidates of a 50-view scene is ~1e8 cell keys once every getFullDepth block is written out per pyramid level -- the
// vectors of Scene::depthFootprints -- and building those took longer than everything the GPU does for the level).
//   wr [n][M][4]: the cell setDepths would write per attached image (view, level, x, y; view < 0: none)
//   fr [n][M][4]: pixelFreeTests' cell per attached image
//   at [n][M][3]: depthTests' 3x3 level-0 pixel block per attached image (view, ix0, iy0)
//   vb [n][V][3]: viewBlockTest's block per view of the scene (examined?, ix0, iy0)
struct RawFootprints {
    size_t n = 0, M = 0, V = 0;
    PVec<int32_t> nimg, wr, fr, at, vb;
    RawFootprints() : nimg(PinnedAlloc<int32_t>(true)), wr(PinnedAlloc<int32_t>(true)), fr(PinnedAlloc<int32_t>(true)),
                      at(PinnedAlloc<int32_t>(true)), vb(PinnedAlloc<int32_t>(true)) {}
};
bool raw_footprints(const Scene& sc, const Patch3d* const* patches, size_t n, RawFootprints& F) {
    hpmvs_scene* dev = sc.deviceScene();
    if (!dev) return false;
    F.n = n; F.V = sc.cameras_.size(); F.M = 1;
    if (n == 0) return true;
    HostBatch hb(patches, n);
    F.M = (size_t)hb.b.max_images;
    F.nimg.assign(hb.nimg.begin(), hb.nimg.end());
    F.wr.resize(n * F.M * 4); F.fr.resize(n * F.M * 4); F.at.resize(n * F.M * 3); F.vb.resize(n * F.V * 3);
    if (hpmvs_depth_footprints_batch(dev, &hb.b, F.wr.data(), F.fr.data(), F.at.data(), F.vb.data(), 0, nullptr) != HPMVS_OK) {
        std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
        return false;
    }
    return true;
}
// The cells Scene::getFullDepth visits for the 3x3 level-0 pixel block from (ix0, iy0) (Scene.cpp:406-432, 538-550): the pixel's
// map cell is pixel / DEPTH_SUBSAMPLE, on level l that cell >> l; pixels with a negative coordinate are never looked up
// (depthTest returns first).  Per level the block's cells are a rectangle of 1 or 2 cells each way: fn(key) for each of them,
// once; stops and returns true as soon as fn does.
template <class Fn>
inline bool for_block_cells(int view, int ix0, int iy0, int nLevels, Fn&& fn) {
    const int x1 = ix0 + 2, y1 = iy0 + 2;
    if (x1 < 0 || y1 < 0) return false;
    const int x0 = ix0 < 0 ? 0 : ix0, y0 = iy0 < 0 ? 0 : iy0;
    for (int l = 0; l < nLevels; l++) {
        const int xs = x0 >> (1 + l), xe = x1 >> (1 + l), ys = y0 >> (1 + l), ye = y1 >> (1 + l);
        for (int y = ys; y <= ye; y++)
            for (int x = xs; x <= xe; x++)
                if (fn(cell_key(view, l, x, y))) return true;
    }
    return false;
}
// every cell candidate i READS (depthTests' and viewBlockTest's blocks on every level, pixelFreeTests' cells)
template <class Fn>
inline bool for_read_cells(const RawFootprints& F, size_t i, int nLevels, Fn&& fn) {
    const size_t m = (size_t)F.nimg[i];
    for (size_t k = 0; k < m; k++) {
        const int32_t* f = &F.fr[(i * F.M + k) * 4]; const int32_t* a = &F.at[(i * F.M + k) * 3];
        if (f[0] >= 0 && fn(cell_key(f[0], f[1], f[2], f[3]))) return true;
        if (a[0] >= 0 && for_block_cells(a[0], a[1], a[2], nLevels, fn)) return true;
    }
    for (size_t v = 0; v < F.V; v++) {
        const int32_t* b = &F.vb[(i * F.V + v) * 3];
        if (b[0] && for_block_cells((int)v, b[1], b[2], nLevels, fn)) return true;
    }
    return false;
}
// every cell setDepths would WRITE for candidate i
template <class Fn>
inline bool for_write_cells(const RawFootprints& F, size_t i, Fn&& fn) {
    const size_t m = (size_t)F.nimg[i];
    for (size_t k = 0; k < m; k++) {
        const int32_t* w = &F.wr[(i * F.M + k) * 4];
        if (w[0] >= 0 && fn(cell_key(w[0], w[1], w[2], w[3]))) return true;
    }
    return false;
}
// a set of 64-bit keys: open addressing, linear probing (the walk of extendLevel does ~1e7 look-ups per wave)
struct FlatSet {
    std::vector<uint64_t> t;
    size_t mask = 0, n = 0;
    FlatSet() { t.assign(1024, 0); mask = 1023; }
    static inline size_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return (size_t)k; }
    void clear() { if (n) { if (t.size() > 4096 && 8 * n < t.size()) { t.assign(t.size() / 2, 0); mask = t.size() - 1; } else std::fill(t.begin(), t.end(), 0); n = 0; } }
    bool empty() const { return n == 0; }
    bool has(uint64_t k) const {
        if (!n) return false;
        k += 1;   // (0 marks an empty slot; no key is all ones)
        for (size_t i = mix(k) & mask;; i = (i + 1) & mask) { const uint64_t v = t[i]; if (v == k) return true; if (!v) return false; }
    }
    void add(uint64_t k) {
        if (2 * (n + 1) > t.size()) grow();
        k += 1;
        for (size_t i = mix(k) & mask;; i = (i + 1) & mask) { if (t[i] == k) return; if (!t[i]) { t[i] = k; n++; return; } }
    }
    void grow() {
        std::vector<uint64_t> old; old.swap(t);
        t.assign(old.size() * 2, 0); mask = t.size() - 1;
        for (uint64_t k : old) if (k) { for (size_t i = mix(k) & mask;; i = (i + 1) & mask) if (!t[i]) { t[i] = k; break; } }
    }
};
// Who can influence whom inside one level, computed ONCE from the footprints (they do not change between the waves):
//   flow[t]: the candidates u < t that would WRITE a cell t reads   (t must wait while such a u is accepted in this wave or undecided)
//   anti[t]: the candidates d < t that READ a cell t would write    (t must not be entered while such a d is undecided)
// A read block (3x3 level-0 pixels, every pyramid level) meets a write (view, l, x, y) iff on level l the block's cell rectangle
// contains (x, y); a pixelFreeTests read meets it iff it is that very cell.  Writes are indexed by (view, 2^S x 2^S level-0 cells)
// buckets -- a write on level l <= S lies in exactly one -- and every read looks at the 1-4 buckets it touches: ~1e7 rectangle
// tests for 1e5 candidates of a 50-view scene, where the sets of cell keys this replaces did ~1e7 hash insertions and look-ups
// per wave.  Indices are positions in the footprint list, which is in candidate (= walk) order.
struct ConflictGraph {
    std::vector<uint32_t> flow_off, flow_adj, anti_off, anti_adj;
};
void build_conflict_graph(const RawFootprints& F, int nLevels, int maxW, int maxH, ConflictGraph& G) {
    const size_t n = F.n;
    const int S = nLevels - 1;
    // bucket size: 2^B x 2^B level-0 cells, B = the level most writes are on (a write on level l <= B lies in one bucket, one on a
    // higher level is entered in the (2^(l-B))^2 buckets it covers), coarser if that would be more than ~16 M buckets
    size_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_writes = 0;
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < (size_t)F.nimg[i]; k++) { const int32_t* w = &F.wr[(i * F.M + k) * 4]; if (w[0] >= 0 && (size_t)w[0] < F.V && w[1] >= 0 && w[1] < nLevels) { hist[w[1] & 7]++; n_writes++; } }
    int B = 1;
    { size_t acc = 0; for (int l = 0; l < nLevels; l++) { acc += hist[l]; if (2 * acc >= n_writes) { B = l; break; } } }
    B = B < 1 ? 1 : (B > S ? S : B);
    auto dims = [&](int b, int& nx, int& ny) { nx = (((maxW + 1) / 2) >> b) + 2; ny = (((maxH + 1) / 2) >> b) + 2; };
    int nbx, nby;
    dims(B, nbx, nby);
    while (B < S && F.V * (size_t)nbx * (size_t)nby > ((size_t)1 << 24)) { B++; dims(B, nbx, nby); }
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    auto bucket = [&](int view, int bx, int by) { return ((size_t)view * (size_t)nby + (size_t)by) * (size_t)nbx + (size_t)bx; };
    struct Wr { uint32_t cand; int32_t l, x, y; };
    const size_t nb = F.V * (size_t)nbx * (size_t)nby;
    std::vector<uint32_t> boff(nb + 1, 0);
    // the buckets a cell (level l, x, y) covers: [bx0, bx1] x [by0, by1] (clamped: the buckets only preselect, the test is exact)
    auto cover = [&](int l, int x, int y, int& bx0, int& bx1, int& by0, int& by1) {
        if (l <= B) { bx0 = bx1 = clampi(x >> (B - l), nbx - 1); by0 = by1 = clampi(y >> (B - l), nby - 1); }
        else {
            const long long sx = (long long)x << (l - B), sy = (long long)y << (l - B), w = ((long long)1 << (l - B)) - 1;
            bx0 = clampi((int)std::min<long long>(sx, nbx), nbx - 1); bx1 = clampi((int)std::min<long long>(sx + w, nbx), nbx - 1);
            by0 = clampi((int)std::min<long long>(sy, nby), nby - 1); by1 = clampi((int)std::min<long long>(sy + w, nby), nby - 1);
        }
    };
    auto each_write = [&](auto&& fn) {
        for (size_t i = 0; i < n; i++)
            for (size_t k = 0; k < (size_t)F.nimg[i]; k++) {
                const int32_t* w = &F.wr[(i * F.M + k) * 4];
                if (w[0] < 0 || (size_t)w[0] >= F.V || w[1] < 0 || w[1] >= nLevels) continue;   // (a level the gates never read)
                int bx0, bx1, by0, by1;
                cover(w[1], w[2], w[3], bx0, bx1, by0, by1);
                for (int by = by0; by <= by1; by++) for (int bx = bx0; bx <= bx1; bx++) fn(bucket(w[0], bx, by), Wr{(uint32_t)i, w[1], w[2], w[3]});
            }
    };
    each_write([&](size_t b, const Wr&) { boff[b + 1]++; });
    for (size_t b = 0; b < nb; b++) boff[b + 1] += boff[b];
    std::vector<Wr> wl(boff[nb]);
    {
        std::vector<uint32_t> fill(boff.begin(), boff.end() - 1);
        each_write([&](size_t b, const Wr& w) { wl[fill[b]++] = w; });
    }
    // the readers, in parallel: every thread collects the edges of its share of the candidates
    std::vector<std::pair<uint32_t, uint32_t> > e_flow, e_anti;   // (node, neighbour)
    const int n_threads = n < 2048 ? 1 : std::min(omp_get_max_threads(), 32);   // (a cgroup quota is usually far below the hardware threads)
#pragma omp parallel num_threads(n_threads)
    {
        std::vector<std::pair<uint32_t, uint32_t> > my_flow, my_anti;
        std::vector<uint32_t> seen(n, 0xFFFFFFFFu);   // seen[writer] == reader: that pair is already in the lists
        auto emit = [&](uint32_t reader, uint32_t writer) {
            if (seen[writer] == reader) return;
            seen[writer] = reader;
            if (writer < reader) my_flow.emplace_back(reader, writer); else my_anti.emplace_back(writer, reader);
        };
        auto block = [&](uint32_t i, int view, int ix0, int iy0) {
            const int x1 = ix0 + 2, y1 = iy0 + 2;
            if (x1 < 0 || y1 < 0 || view < 0 || (size_t)view >= F.V) return;
            const int x0 = ix0 < 0 ? 0 : ix0, y0 = iy0 < 0 ? 0 : iy0;
            const int bxs = clampi((x0 >> 1) >> B, nbx - 1), bxe = clampi((x1 >> 1) >> B, nbx - 1), bys = clampi((y0 >> 1) >> B, nby - 1), bye = clampi((y1 >> 1) >> B, nby - 1);
            for (int by = bys; by <= bye; by++)
                for (int bx = bxs; bx <= bxe; bx++) {
                    const size_t b = bucket(view, bx, by);
                    for (uint32_t q = boff[b]; q < boff[b + 1]; q++) {
                        const Wr& w = wl[q];
                        if (w.cand == i) continue;
                        const int sh = 1 + w.l;
                        if (w.x >= (x0 >> sh) && w.x <= (x1 >> sh) && w.y >= (y0 >> sh) && w.y <= (y1 >> sh)) emit(i, w.cand);
                    }
                }
        };
#pragma omp for schedule(dynamic, 256) nowait
        for (long long ii = 0; ii < (long long)n; ii++) {
            const size_t i = (size_t)ii;
            const size_t m = (size_t)F.nimg[i];
            for (size_t k = 0; k < m; k++) {
                const int32_t* f = &F.fr[(i * F.M + k) * 4]; const int32_t* a = &F.at[(i * F.M + k) * 3];
                if (f[0] >= 0 && (size_t)f[0] < F.V && f[1] >= 0 && f[1] < nLevels) {   // pixelFreeTests: one cell on its own level
                    int bx0, bx1, by0, by1;
                    cover(f[1], f[2], f[3], bx0, bx1, by0, by1);
                    const size_t b = bucket(f[0], bx0, by0);   // (any bucket the cell covers holds every write of that very cell)
                    for (uint32_t q = boff[b]; q < boff[b + 1]; q++) { const Wr& w = wl[q]; if (w.cand != (uint32_t)i && w.l == f[1] && w.x == f[2] && w.y == f[3]) emit((uint32_t)i, w.cand); }
                }
                if (a[0] >= 0) block((uint32_t)i, a[0], a[1], a[2]);
            }
            for (size_t v = 0; v < F.V; v++) {
                const int32_t* b = &F.vb[(i * F.V + v) * 3];
                if (b[0]) block((uint32_t)i, (int)v, b[1], b[2]);
            }
        }
#pragma omp critical
        {
            e_flow.insert(e_flow.end(), my_flow.begin(), my_flow.end());
            e_anti.insert(e_anti.end(), my_anti.begin(), my_anti.end());
        }
    }
    auto csr = [&](std::vector<std::pair<uint32_t, uint32_t> >& e, std::vector<uint32_t>& off, std::vector<uint32_t>& adj) {
        off.assign(n + 1, 0);
        for (auto& p : e) off[p.first + 1]++;
        for (size_t i = 0; i < n; i++) off[i + 1] += off[i];
        adj.resize(e.size());
        std::vector<uint32_t> fill(off.begin(), off.end() - 1);
        for (auto& p : e) adj[fill[p.first]++] = p.second;
    };
    csr(e_flow, G.flow_off, G.flow_adj);
    csr(e_anti, G.anti_off, G.anti_adj);
}
}  // namespace

// Test hook (tests/test_cpu_host_logic.py; no device needed): random footprints -- blocks at the image border and beyond it,
// writes on every pyramid level, cells that collide -- and the conflict graph of build_conflict_graph against the definition
// itself: candidate j is a neighbour of i iff a cell key j would write is among the cell keys i reads (the vectors of
// Scene::depthFootprints).  Returns the number of (i, j) pairs on which the two disagree (0), or -1 for bad arguments.
extern "C" int hpmvs_host_selftest_conflict_graph(unsigned seed, int n, int n_views, int max_w, int max_h, int n_levels, int max_images, int* n_edges) {
    if (n < 1 || n_views < 1 || max_w < 8 || max_h < 8 || n_levels < 1 || n_levels > 7 || max_images < 1) return -1;
    RawFootprints F;
    F.n = (size_t)n; F.V = (size_t)n_views; F.M = (size_t)max_images;
    F.nimg.assign(F.n, 0);
    F.wr.assign(F.n * F.M * 4, -1); F.fr.assign(F.n * F.M * 4, -1); F.at.assign(F.n * F.M * 3, -1); F.vb.assign(F.n * F.V * 3, 0);
    unsigned long long st = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)seed * 0xD1B54A32D192ED03ull);
    auto rnd = [&](int m) { st = st * 6364136223846793005ull + 1442695040888963407ull; return (int)((st >> 33) % (unsigned long long)m); };
    // everything happens in a few small windows of the images, so that candidates really meet
    const int wx = max_w / 4 < 24 ? max_w / 4 : 24, wy = max_h / 4 < 24 ? max_h / 4 : 24;
    for (size_t i = 0; i < F.n; i++) {
        const int m = 1 + rnd(max_images);
        F.nimg[i] = m;
        const int corner = rnd(4);   // windows at the four corners: blocks reach outside the image there
        const int bx = (corner & 1) ? max_w - wx - 1 : -2, by = (corner & 2) ? max_h - wy - 1 : -2;
        for (int k = 0; k < m; k++) {
            const int view = rnd(n_views), px = bx + rnd(wx + 3), py = by + rnd(wy + 3);
            int32_t* a = &F.at[(i * F.M + k) * 3];
            if (rnd(8)) { a[0] = view; a[1] = px - 1; a[2] = py - 1; }
            const int l = rnd(n_levels);
            const int cx = (px < 0 ? 0 : px) >> (1 + l), cy = (py < 0 ? 0 : py) >> (1 + l);
            int32_t* w = &F.wr[(i * F.M + k) * 4];
            if (rnd(6)) { w[0] = view; w[1] = l; w[2] = cx + rnd(2); w[3] = cy + rnd(2); }
            const int lf = rnd(n_levels);
            int32_t* f = &F.fr[(i * F.M + k) * 4];
            if (rnd(6)) { f[0] = view; f[1] = lf; f[2] = ((px < 0 ? 0 : px) >> (1 + lf)) + rnd(2); f[3] = ((py < 0 ? 0 : py) >> (1 + lf)) + rnd(2); }
        }
        for (int v = 0; v < n_views; v++)
            if (rnd(3) == 0) { int32_t* b = &F.vb[(i * F.V + (size_t)v) * 3]; b[0] = 1; b[1] = bx + rnd(wx + 3) - 1; b[2] = by + rnd(wy + 3) - 1; }
    }
    ConflictGraph G;
    build_conflict_graph(F, n_levels, max_w, max_h, G);
    // the definition: cell key -> the candidates that would write it
    std::unordered_map<uint64_t, std::vector<uint32_t> > writers;
    for (size_t j = 0; j < F.n; j++) for_write_cells(F, j, [&](uint64_t k) { writers[k].push_back((uint32_t)j); return false; });
    int bad = 0;
    for (size_t i = 0; i < F.n; i++) {
        std::unordered_set<uint32_t> want_flow, want_anti_of;   // j < i that write what i reads; j > i that write what i reads (i is in anti[j])
        for_read_cells(F, i, n_levels, [&](uint64_t k) {
            auto it = writers.find(k);
            if (it != writers.end()) for (uint32_t j : it->second) { if (j < i) want_flow.insert(j); else if (j > i) want_anti_of.insert(j); }
            return false;
        });
        std::unordered_set<uint32_t> got_flow(G.flow_adj.begin() + G.flow_off[i], G.flow_adj.begin() + G.flow_off[i + 1]);
        if (got_flow != want_flow || got_flow.size() != (size_t)(G.flow_off[i + 1] - G.flow_off[i])) bad++;
        for (uint32_t j : want_anti_of) {
            bool found = false;
            for (uint32_t q = G.anti_off[j]; q < G.anti_off[j + 1]; q++) found |= G.anti_adj[q] == (uint32_t)i;
            if (!found) bad++;
        }
    }
    // ... and no anti edge that the definition does not have
    size_t want_anti_total = 0;
    for (size_t i = 0; i < F.n; i++) {
        std::unordered_set<uint32_t> later;
        for_read_cells(F, i, n_levels, [&](uint64_t k) {
            auto it = writers.find(k);
            if (it != writers.end()) for (uint32_t j : it->second) if (j > i) later.insert(j);
            return false;
        });
        want_anti_total += later.size();
    }
    if (want_anti_total != G.anti_adj.size()) bad++;
    if (n_edges) *n_edges = (int)(G.flow_adj.size() + G.anti_adj.size());
    return bad;
}

bool Scene::depthFootprints(const Patch3d* const* patches, size_t n, std::vector<std::vector<uint64_t> >& reads,
                            std::vector<std::vector<uint64_t> >& writes, int nLevels) const {
    reads.assign(n, std::vector<uint64_t>());
    writes.assign(n, std::vector<uint64_t>());
    RawFootprints F;
    if (!raw_footprints(*this, patches, n, F)) return false;
    for (size_t i = 0; i < n; i++) {
        for_read_cells(F, i, nLevels, [&](uint64_t k) { reads[i].push_back(k); return false; });
        for_write_cells(F, i, [&](uint64_t k) { writes[i].push_back(k); return false; });
    }
    return true;
}

// ---------------------------------------------------------------- PatchOptimizer
PatchOptimizer::PatchOptimizer(const mo3d::HpmvsOptions& options, const mo3d::Scene* scene)
    : options_p(&options), scene_p(scene) {}

bool PatchOptimizer::optimize(mo3d::Patch3d& patch) {
    // Default: a batch of one through the C ABI, which serves small host batches from ONE long-running launch that all
    // calling threads feed (capi.hip, "open batch"): a caller waits for its own patch only.  HPMVS_SERVICE=0 selects
    // the round-2 combiner below (concurrent callers are grouped into launches and wait for the group's slowest patch).
    static const bool use_service = [] { const char* e = getenv("HPMVS_SERVICE"); return !(e && e[0] == '0'); }();
    if (use_service) {
        mo3d::Patch3d* one = &patch;
        uint8_t ok = 0;
        optimizeBatch(&one, 1, &ok);
        return ok != 0;
    }
    Combiner& C = *(Combiner*)scene_p->combiner();
    Combiner::Entry me;
    me.patch = &patch; me.opt = options_p;
    std::unique_lock<std::mutex> lk(C.mu);
    C.pending.push_back(&me);
    C.inside++;
    C.peak = std::max((double)C.inside, 0.98 * C.peak);
    if (C.collecting) {
        C.cv_collect.notify_one();
        C.cv_done.wait(lk, [&] { return me.done; });
    } else {
        C.collecting = true;
        // Collect: a lone caller launches at once.  With several threads calling, give the others a short window
        // to join (a lone patch needs milliseconds on the device, the window is a fraction of that), and while
        // kMaxCombinedLaunches batches are already on the device keep collecting until one of them returns.
        if ((int)(C.peak + 0.5) > 1) {
            const auto t_open = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
            for (;;) {
                const bool everyone = (int)C.pending.size() >= (int)(C.peak + 0.5);
                const bool window_over = std::chrono::steady_clock::now() >= t_open;
                if (C.launches < kMaxCombinedLaunches && (everyone || window_over)) break;
                C.cv_collect.wait_for(lk, std::chrono::microseconds(100));
            }
        }
        std::vector<Combiner::Entry*> take;
        take.swap(C.pending);
        C.collecting = false;
        C.launches++;
        lk.unlock();
        // Whatever happens in the launch (an exception from a vector allocation, an infrastructure error), the
        // owners of the taken entries are woken and the launch slot is given back: entries that were
        // not served keep ok = 0, i.e. "patch untouched".
        struct Finish {
            Combiner& C; std::vector<Combiner::Entry*>& take; std::unique_lock<std::mutex>& lk;
            ~Finish() {
                if (!lk.owns_lock()) lk.lock();
                C.launches--;
                for (Combiner::Entry* e : take) e->done = true;
                C.cv_done.notify_all();
                C.cv_collect.notify_one();  // a collector may be waiting for a launch slot
            }
        } finish{C, take, lk};
        // one launch per distinct option SET (normally one; compared by value)
        std::vector<char> handled(take.size(), 0);
        for (size_t a = 0; a < take.size(); a++) {
            if (handled[a]) continue;
            std::vector<Patch3d*> group;
            std::vector<size_t> who;
            for (size_t k = a; k < take.size(); k++)
                if (!handled[k] && (take[k]->opt == take[a]->opt || same_options(*take[k]->opt, *take[a]->opt))) {
                    group.push_back(take[k]->patch); who.push_back(k); handled[k] = 1;
                }
            std::vector<uint8_t> okv(group.size(), 0);
            PatchOptimizer runner(*take[a]->opt, scene_p);
            runner.optimizeBatch(group.data(), group.size(), okv.data());
            for (size_t k = 0; k < who.size(); k++) {
                take[who[k]]->ok = okv[k];
                take[who[k]]->f = runner.lastF_.size() > k ? runner.lastF_[k] : 0.0;
                take[who[k]]->evals = runner.lastEvals_.size() > k ? runner.lastEvals_[k] : 0;
            }
        }
    }
    if (!lk.owns_lock()) lk.lock();
    C.inside--;
    lastF_.assign(1, me.f);
    lastEvals_.assign(1, me.evals);
    return me.ok != 0;
}

size_t PatchOptimizer::pinnedCacheBytes() { return tl_pins.cached_bytes(); }
size_t PatchOptimizer::pinnedCacheCap() { return tl_pins.cap_; }
void PatchOptimizer::setPinnedCacheCap(size_t bytes) { tl_pins.set_cap(bytes); }

size_t PatchOptimizer::optimizeBatch(mo3d::Patch3d* const* patches, size_t n, uint8_t* ok) {
    lastF_.assign(n, 0.0);
    lastEvals_.assign(n, 0);
    for (size_t i = 0; i < n; i++) ok[i] = 0;
    if (n == 0) return 0;
    hpmvs_scene* dev = scene_p->deviceScene();
    if (!dev) return 0;
    size_t longest_in = 0;
    for (size_t i = 0; i < n; i++) longest_in = std::max(longest_in, patches[i]->images_.size());
    // (tl_full_rows: the retry below -- rows as wide as a list can get, whatever the batch size)
    const int full = list_stride(scene_p->cameras_.size(), (size_t)-1, longest_in);
    // A call of up to four patches takes the open batch's 64-id rows first and is refined again with full rows when a list outgrows
    // them (below).  In a dense scene nearly every patch does: after such a retry the next small calls of this optimizer go straight
    // to full rows (a batch launch of their own, whose wide kernel takes the long lists) instead of failing twice first; the result is
    // the same patch either way.  Every 16th small call tries the cheap way again.
    const bool straight = !tl_full_rows && n <= 4 && preferFullRows_ > 0;
    if (straight) preferFullRows_--;
    const int stride = (tl_full_rows || straight) ? full : list_stride(scene_p->cameras_.size(), n, longest_in);
    const PinnedAlloc<char> pin(n > kPinFrom);
    PVec<float> center(4 * n, pin), normal(4 * n, pin), scale(n, pin), color(3 * n, pin), ncc(n, pin);
    PVec<int32_t> nimg(n, pin), images(n * stride, -1, pin), evals(n, pin), stage(n, pin);
    PVec<double> fmin(n, pin);
    PVec<uint8_t> okp(n, pin);  // (the caller's ok[] is ordinary memory: a copy here keeps the call free of device allocations)
    for (size_t i = 0; i < n; i++) {
        const Patch3d& p = *patches[i];
        for (int k = 0; k < 4; k++) { center[4 * i + k] = p.center_[k]; normal[4 * i + k] = p.normal_[k]; }
        scale[i] = p.scale_3dx_;
        // more attached images than the device cap: reported as a failed patch (stage 100)
        nimg[i] = (int32_t)p.images_.size();
        for (size_t k = 0; k < p.images_.size() && k < (size_t)stride; k++) images[i * stride + k] = p.images_[k];
    }
    hpmvs_options o;
    o.MAXLEVEL = options_p->MAXLEVEL; o.MINLEVEL = options_p->MINLEVEL;
    o.MAX_ANGLE = options_p->MAX_ANGLE; o.MIN_ANGLE = options_p->MIN_ANGLE;
    o.MAX_IMAGES_PER_PATCH = options_p->MAX_IMAGES_PER_PATCH; o.MIN_IMAGES_PER_PATCH = options_p->MIN_IMAGES_PER_PATCH;
    o.NCC_ALPHA_1 = options_p->NCC_ALPHA_1; o.NCC_ALPHA_2 = options_p->NCC_ALPHA_2;
    hpmvs_patch_batch b;
    memset(&b, 0, sizeof(b));
    b.n = (int32_t)n; b.max_images = stride;
    b.center = center.data(); b.normal = normal.data(); b.scale = scale.data();
    b.n_images = nimg.data(); b.images = images.data();
    b.ok = okp.data(); b.color = color.data(); b.ncc = ncc.data(); b.fmin = fmin.data(); b.nevals = evals.data();
    b.stage = stage.data();
    if (hpmvs_optimize_batch(dev, &o, &b, 0, nullptr) != HPMVS_OK) {
        std::cerr << "patch optimization failed: <" << hpmvs_last_error() << ">" << std::endl;
        for (size_t i = 0; i < n; i++) ok[i] = 0;
        return 0;
    }
    memcpy(ok, okp.data(), n);
    {   // HPMVS_DEBUG_CALLS=1: one line per call of up to four patches (what came back, per attempt)
        static const bool dbg = [] { const char* e = getenv("HPMVS_DEBUG_CALLS"); return e && e[0] == '1'; }();
        if (dbg && n <= 4)
            for (size_t i = 0; i < n; i++)
                std::fprintf(stderr, "hpmvs call: in %.9g %.9g %.9g (%zu ids) rows %d full %d -> ok %d stage %d nimg %d evals %d out %.9g %.9g %.9g\n",
                             patches[i]->center_[0], patches[i]->center_[1], patches[i]->center_[2], patches[i]->images_.size(), stride, (int)tl_full_rows,
                             (int)ok[i], stage[i], nimg[i], evals[i], center[4 * i], center[4 * i + 1], center[4 * i + 2]);
    }
    // The rows of a call of up to four patches are the open batch's 64 ids.  A patch that ends at stage 100 there (its refined
    // list is longer than that: a dense scene) was left untouched; the reference's list is unbounded, so it is refined again
    // with rows as wide as a list can get.
    std::vector<size_t> again;
    if (stride < full)
        for (size_t i = 0; i < n; i++) if (!ok[i] && stage[i] == 100) again.push_back(i);
    size_t good = 0;
    for (size_t i = 0; i < n; i++) {
        lastF_[i] = fmin[i];
        lastEvals_[i] = evals[i];
        if (!ok[i]) continue;  // patch untouched on failure (reference PatchOptimizer.cpp:86-87)
        Patch3d& p = *patches[i];
        for (int k = 0; k < 4; k++) { p.center_[k] = center[4 * i + k]; p.normal_[k] = normal[4 * i + k]; }
        p.scale_3dx_ = scale[i];
        p.images_.assign(images.begin() + i * stride, images.begin() + i * stride + nimg[i]);
        p.ncc_ = ncc[i];
        p.color_ = Eigen::Vector3f(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
        good++;
    }
    if (!again.empty()) {
        preferFullRows_ = 16;
        std::vector<mo3d::Patch3d*> sub(again.size());
        std::vector<uint8_t> sok(again.size(), 0);
        for (size_t k = 0; k < again.size(); k++) sub[k] = patches[again[k]];
        const std::vector<double> f0 = lastF_;
        const std::vector<int> e0 = lastEvals_;
        tl_full_rows = true;
        good += optimizeBatch(sub.data(), sub.size(), sok.data());
        tl_full_rows = false;
        const std::vector<double> f1 = lastF_;
        const std::vector<int> e1 = lastEvals_;
        lastF_ = f0; lastEvals_ = e0;
        for (size_t k = 0; k < again.size(); k++) { ok[again[k]] = sok[k]; lastF_[again[k]] = f1[k]; lastEvals_[again[k]] = e1[k]; }
    }
    return good;
}

int PatchOptimizer::expandBatch(ExpandMode mode, const mo3d::Patch3d* const* parents, const CellRef* cells, size_t n,
                                const uint8_t* skip, std::vector<mo3d::Patch3d>& candidates,
                                std::vector<uint8_t>& accepted) {
    const int N = (mode == EXTEND) ? 6 : 4;
    candidates.clear();
    accepted.assign(n * N, 0);
    lastF_.assign(n * N, 0.0);
    lastEvals_.assign(n * N, 0);
    if (n == 0) return N;
    hpmvs_scene* dev = scene_p->deviceScene();
    if (!dev) return N;
    size_t longest_in = 0;
    for (size_t i = 0; i < n; i++) longest_in = std::max(longest_in, parents[i]->images_.size());
    const int stride = list_stride(scene_p->cameras_.size(), n * N, longest_in);
    const size_t m = n * N;
    const PinnedAlloc<char> pin(true);  // (this entry stages whatever the batch size: pinned arrays are used in place)
    PVec<float> pc(4 * n, pin), pn(4 * n, pin), ps(n, pin), cc(3 * n, pin), cw(n, pin);
    PVec<int32_t> pnimg(n, pin), pimg(n * stride, -1, pin);
    for (size_t i = 0; i < n; i++) {
        const Patch3d& p = *parents[i];
        for (int k = 0; k < 4; k++) { pc[4 * i + k] = p.center_[k]; pn[4 * i + k] = p.normal_[k]; }
        ps[i] = p.scale_3dx_;
        pnimg[i] = (int32_t)p.images_.size();
        for (size_t k = 0; k < p.images_.size() && k < (size_t)stride; k++) pimg[i * stride + k] = p.images_[k];
        for (int k = 0; k < 3; k++) cc[3 * i + k] = cells[i].c[k];
        cw[i] = cells[i].width;
    }
    PVec<float> center(4 * m, pin), normal(4 * m, pin), scale(m, pin), color(3 * m, pin), ncc(m, pin);
    PVec<int32_t> nimg(m, pin), images(m * stride, -1, pin), evals(m, pin);
    PVec<double> fmin(m, pin);
    hpmvs_options o;
    o.MAXLEVEL = options_p->MAXLEVEL; o.MINLEVEL = options_p->MINLEVEL;
    o.MAX_ANGLE = options_p->MAX_ANGLE; o.MIN_ANGLE = options_p->MIN_ANGLE;
    o.MAX_IMAGES_PER_PATCH = options_p->MAX_IMAGES_PER_PATCH; o.MIN_IMAGES_PER_PATCH = options_p->MIN_IMAGES_PER_PATCH;
    o.NCC_ALPHA_1 = options_p->NCC_ALPHA_1; o.NCC_ALPHA_2 = options_p->NCC_ALPHA_2;
    hpmvs_patch_batch pb, ob;
    memset(&pb, 0, sizeof(pb));
    memset(&ob, 0, sizeof(ob));
    pb.n = (int32_t)n; pb.max_images = stride;
    pb.center = pc.data(); pb.normal = pn.data(); pb.scale = ps.data(); pb.n_images = pnimg.data(); pb.images = pimg.data();
    ob.n = (int32_t)m; ob.max_images = stride;
    ob.center = center.data(); ob.normal = normal.data(); ob.scale = scale.data(); ob.n_images = nimg.data();
    PVec<uint8_t> accp(m, pin), skp(skip ? m : 0, pin);  // (pinned copies of the caller's ordinary arrays: no device allocation in the call)
    if (skip) memcpy(skp.data(), skip, m);
    ob.images = images.data(); ob.ok = accp.data(); ob.color = color.data(); ob.ncc = ncc.data();
    ob.fmin = fmin.data(); ob.nevals = evals.data();
    if (hpmvs_expand_batch(dev, &o, (int)mode, &pb, cc.data(), cw.data(), skip ? skp.data() : nullptr, &ob, 0, nullptr) != HPMVS_OK) {
        std::cerr << "frontier expansion failed: <" << hpmvs_last_error() << ">" << std::endl;
        accepted.assign(m, 0);
        return N;
    }
    memcpy(accepted.data(), accp.data(), m);
    candidates.resize(m);
    // (a frontier of 1e5 candidates: 1e5 copies of a parent with its image vector -- in parallel from 4096 candidates on)
#pragma omp parallel for schedule(static) num_threads(m < 4096 ? 1 : std::min(omp_get_max_threads(), 16))
    for (long long tt = 0; tt < (long long)m; tt++) {
        const size_t t = (size_t)tt;
        Patch3d& q = candidates[t];
        q = *parents[t / N];  // *newP = *p (CellProcessor.cpp:113, 238)
        q.expanded_ = false;
        q.flatness_ = -1.0f;
        for (int k = 0; k < 4; k++) { q.center_[k] = center[4 * t + k]; q.normal_[k] = normal[4 * t + k]; }
        q.scale_3dx_ = scale[t];
        lastF_[t] = fmin[t];
        lastEvals_[t] = evals[t];
        if (!accepted[t] && nimg[t] <= 0) continue;  // never optimized: keeps the parent's image list
        if (accepted[t]) {
            q.images_.assign(images.begin() + t * stride, images.begin() + t * stride + nimg[t]);
            q.ncc_ = ncc[t];
            q.color_ = Eigen::Vector3f(color[3 * t], color[3 * t + 1], color[3 * t + 2]);
        }
    }
    return N;
}

bool PatchOptimizer::branchLevel(const mo3d::Patch3d* const* parents, const CellRef* cells, size_t n, const uint8_t* finalLevel,
                                 BranchResult& R) {
    R = BranchResult();
    if (n == 0) return true;
    if (!scene_p->levelSupport(parents, n, options_p->MINLEVEL, R.support)) return false;
    std::vector<uint8_t> skip(4 * n, 0), acc;
    for (size_t i = 0; i < n; i++) if (R.support[i] < 1) for (int k = 0; k < 4; k++) skip[4 * i + k] = 1;   // exhausted: builds nothing
    expandBatch(BRANCH, parents, cells, n, skip.data(), R.candidates, acc);
    if (R.candidates.size() != 4 * n) return false;
    R.child.assign(4 * n, 0);
    R.split.assign(n, 0);
    std::vector<const Patch3d*> ops;
    std::vector<uint8_t> sub;
    for (size_t i = 0; i < n; i++) {
        int children = 0;
        for (int k = 0; k < 4; k++) { R.child[4 * i + k] = (acc[4 * i + k] && !skip[4 * i + k]) ? 1 : 0; children += R.child[4 * i + k]; }
        if (R.support[i] < 1 || (finalLevel && finalLevel[i] && children == 0)) continue;
        R.split[i] = 1;
        ops.push_back(parents[i]); sub.push_back(1);                       // the old patch out (CellProcessor.cpp:276-279)
        for (int k = 0; k < 4; k++) if (R.child[4 * i + k]) { ops.push_back(&R.candidates[4 * i + k]); sub.push_back(0); }   // the children in (:296)
    }
    return ops.empty() || scene_p->setDepths(ops.data(), ops.size(), sub.data());
}

// The centres of a level's candidates BEFORE optimize (CellProcessor.cpp:103-116: the leaf look-up of :118-122 uses them): the device's
// own construction with every candidate skipped, read back without building 6 n Patch3d objects, in rows just wide enough for the
// parents' lists (nothing is refined, no list grows).  centers: 3 floats per candidate.
static bool candidate_centers(const Scene* scene, const HpmvsOptions* opt, int mode, const mo3d::Patch3d* const* parents, const float* widths,
                              size_t n, std::vector<float>& centers) {
    const size_t N = mode == 0 ? 6 : 4, m = n * N;
    centers.assign(3 * m, 0.0f);
    if (n == 0) return true;
    hpmvs_scene* dev = scene->deviceScene();
    if (!dev) return false;
    size_t stride = 1;
    for (size_t i = 0; i < n; i++) stride = std::max(stride, parents[i]->images_.size());
    if (stride > (size_t)HPMVS_MAX_IMAGES) stride = HPMVS_MAX_IMAGES;
    const PinnedAlloc<char> pin(true);
    PVec<float> pc(4 * n, pin), pn(4 * n, pin), ps(n, pin), cc(3 * n, 0.0f, pin), cw(widths, widths + n, pin);
    PVec<int32_t> pnimg(n, pin), pimg(n * stride, -1, pin);
    for (size_t i = 0; i < n; i++) {
        const Patch3d& p = *parents[i];
        for (int k = 0; k < 4; k++) { pc[4 * i + k] = p.center_[k]; pn[4 * i + k] = p.normal_[k]; }
        ps[i] = p.scale_3dx_;
        const size_t mi = std::min(p.images_.size(), stride);
        pnimg[i] = (int32_t)mi;
        for (size_t k = 0; k < mi; k++) pimg[i * stride + k] = p.images_[k];
    }
    PVec<float> center(4 * m, pin), normal(4 * m, pin), scale(m, pin);
    PVec<int32_t> nimg(m, pin), images(m * stride, pin);
    PVec<uint8_t> okp(m, pin), skp(m, (uint8_t)1, pin);
    hpmvs_options o;
    o.MAXLEVEL = opt->MAXLEVEL; o.MINLEVEL = opt->MINLEVEL; o.MAX_ANGLE = opt->MAX_ANGLE; o.MIN_ANGLE = opt->MIN_ANGLE;
    o.MAX_IMAGES_PER_PATCH = opt->MAX_IMAGES_PER_PATCH; o.MIN_IMAGES_PER_PATCH = opt->MIN_IMAGES_PER_PATCH;
    o.NCC_ALPHA_1 = opt->NCC_ALPHA_1; o.NCC_ALPHA_2 = opt->NCC_ALPHA_2;
    hpmvs_patch_batch pb, ob;
    memset(&pb, 0, sizeof(pb)); memset(&ob, 0, sizeof(ob));
    pb.n = (int32_t)n; pb.max_images = (int32_t)stride;
    pb.center = pc.data(); pb.normal = pn.data(); pb.scale = ps.data(); pb.n_images = pnimg.data(); pb.images = pimg.data();
    ob.n = (int32_t)m; ob.max_images = (int32_t)stride;
    ob.center = center.data(); ob.normal = normal.data(); ob.scale = scale.data(); ob.n_images = nimg.data(); ob.images = images.data(); ob.ok = okp.data();
    if (hpmvs_expand_batch(dev, &o, mode, &pb, cc.data(), cw.data(), skp.data(), &ob, 0, nullptr) != HPMVS_OK) {
        std::cerr << "hpmvs: " << hpmvs_last_error() << std::endl;
        return false;
    }
    for (size_t t = 0; t < m; t++) for (int k = 0; k < 3; k++) centers[3 * t + k] = center[4 * t + k];
    return true;
}

// HPMVS_LEVEL_TIMES=1: extendLevel prints where its time went (stderr)
static bool level_times() { static const bool on = [] { const char* e = getenv("HPMVS_LEVEL_TIMES"); return e && e[0] == '1'; }(); return on; }
static double level_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t grid_leaf_key(const Eigen::Vector3f& p, float width, void*) {
    const long long ix = (long long)std::floor(p[0] / width), iy = (long long)std::floor(p[1] / width), iz = (long long)std::floor(p[2] / width);
    return (uint64_t)(((ix + (1 << 20)) << 42) | ((iy + (1 << 20)) << 21) | (iz + (1 << 20)));
}

bool PatchOptimizer::extendLevel(const mo3d::Patch3d* const* parents, size_t n, float width, std::unordered_set<uint64_t>& occupied,
                                 float margin, bool absInt, LevelResult& R, bool sequential, LeafKeyFn leafKey, void* user) {
    if (!leafKey) leafKey = grid_leaf_key;
    const size_t N = 6, T = n * N;
    R = LevelResult();
    if (n == 0) return true;
    const double t_begin = level_now();
    std::vector<CellRef> cells(n);
    for (size_t i = 0; i < n; i++) { cells[i].c = Eigen::Vector3f(0.0f, 0.0f, 0.0f); cells[i].width = width; }
    // the candidates before optimize (everything skipped: constructed only) -> their leaves
    std::vector<uint8_t> skip(T, 0), refined;
    std::vector<float> pre, widths(n, width);
    if (!candidate_centers(scene_p, options_p, (int)EXTEND, parents, widths.data(), n, pre)) return false;
    std::vector<uint64_t> preKey(T), postKey(T, 0);
    for (size_t t = 0; t < T; t++) {
        preKey[t] = leafKey(Eigen::Vector3f(pre[3 * t], pre[3 * t + 1], pre[3 * t + 2]), width, user);
        skip[t] = occupied.count(preKey[t]) ? 1 : 0;   // level-start occupancy: those are never refined
    }
    const double t_pre = level_now();
    expandBatch(EXTEND, parents, cells.data(), n, skip.data(), R.candidates, refined);
    if (R.candidates.size() != T) return false;
    const double t_ref = level_now();
    std::vector<const Patch3d*> ptr(T);
    for (size_t t = 0; t < T; t++) {
        ptr[t] = &R.candidates[t];
        if (skip[t]) refined[t] = 0;
        if (refined[t]) postKey[t] = leafKey(Eigen::Vector3f(R.candidates[t].center_[0], R.candidates[t].center_[1], R.candidates[t].center_[2]), width, user);
    }
    // footprints of the REFINED candidates only (the others read and write no map), as the device returns them
    const int nLevels = 6;
    std::vector<const Patch3d*> rptr;
    std::vector<int> fpi(T, -1);
    for (size_t t = 0; t < T; t++) if (refined[t]) { fpi[t] = (int)rptr.size(); rptr.push_back(ptr[t]); }
    RawFootprints F;
    if (!raw_footprints(*scene_p, rptr.data(), rptr.size(), F)) return false;
    const double t_fp = level_now();
    R.stage.assign(T, 1);
    R.counts.assign(3 * T, -1);
    std::vector<size_t> pending;
    for (size_t t = 0; t < T; t++) { if (skip[t]) R.stage[t] = 20; else pending.push_back(t); }
    const int MIN = options_p->MIN_IMAGES_PER_PATCH;
    // the level's conflict graph (who writes what whom reads): once, from the footprints
    ConflictGraph G;
    {
        int maxW = 1, maxH = 1;
        for (const Image& im : scene_p->images_) { maxW = std::max(maxW, im.getWidth()); maxH = std::max(maxH, im.getHeight()); }
        if (sequential && F.n > 0) build_conflict_graph(F, nLevels, maxW, maxH, G);   // (the plain frontier round decides everything in one wave)
    }
    const double t_graph = level_now();
    // open[i] == w: candidate i (footprint index) was accepted or deferred in wave w -- its writes are not in the maps this wave's
    // counts were read from (`dirty`); defer_w[i] == w: it was deferred in wave w -- nobody may overwrite what it reads (`guard`)
    std::vector<int> open(rptr.size(), 0), defer_w(rptr.size(), 0);
    FlatSet maybeOcc, occGuard;
    auto reads_hit_dirty = [&](size_t i, int wave) {
        for (uint32_t q = G.flow_off[i]; q < G.flow_off[i + 1]; q++) if (open[G.flow_adj[q]] == wave) return true;
        return false;
    };
    auto writes_hit_guard = [&](size_t i, int wave) {
        for (uint32_t q = G.anti_off[i]; q < G.anti_off[i + 1]; q++) if (defer_w[G.anti_adj[q]] == wave) return true;
        return false;
    };
    double t_gates = 0.0, t_walk = 0.0, t_set = 0.0;
    while (!pending.empty()) {
        R.waves++;
        double w0 = level_now();
        std::vector<size_t> todo;
        for (size_t t : pending) if (refined[t]) todo.push_back(t);
        std::vector<int> v, b, f;
        std::vector<const Patch3d*> sub(todo.size());
        for (size_t k = 0; k < todo.size(); k++) sub[k] = ptr[todo[k]];
        if (!todo.empty() && !scene_p->depthGates(sub.data(), sub.size(), margin, v, b, f, absInt)) return false;
        std::vector<int> slot(T, -1);
        for (size_t k = 0; k < todo.size(); k++) slot[todo[k]] = (int)k;
        double w1 = level_now();
        t_gates += w1 - w0;
        maybeOcc.clear(); occGuard.clear();
        const int wave = R.waves;
        std::vector<size_t> deferred, accNow;
        auto defer = [&](size_t t) {
            deferred.push_back(t);
            occGuard.add(preKey[t]);
            if (refined[t]) {
                open[(size_t)fpi[t]] = wave; defer_w[(size_t)fpi[t]] = wave;
                maybeOcc.add(postKey[t]); occGuard.add(postKey[t]);
            }
        };
        for (size_t t : pending) {
            if (occupied.count(preKey[t])) { R.stage[t] = 20; continue; }   // its leaf was taken by an earlier candidate
            if (sequential && maybeOcc.has(preKey[t])) { defer(t); continue; }
            if (!refined[t]) { R.stage[t] = 1; continue; }                  // failed in optimize or at the scale / drift gates
            if (sequential && reads_hit_dirty((size_t)fpi[t], wave)) { defer(t); continue; }
            const int k = slot[t];
            const int cv = v[k], cb = b[k], cf = f[k];
            R.counts[3 * t] = cv; R.counts[3 * t + 1] = cb; R.counts[3 * t + 2] = cf;
            if (!(cv >= MIN)) R.stage[t] = 23;
            else if (!(cb < MIN)) R.stage[t] = 24;
            else if (!(cf >= MIN - 1 && cf * 1.0 / (double)R.candidates[t].images_.size() > 0.75)) R.stage[t] = 25;
            else if (occupied.count(postKey[t])) R.stage[t] = 26;
            else if (sequential && (maybeOcc.has(postKey[t]) || occGuard.has(postKey[t]) || writes_hit_guard((size_t)fpi[t], wave))) {
                R.counts[3 * t] = R.counts[3 * t + 1] = R.counts[3 * t + 2] = -1;
                defer(t);
            } else {
                occupied.insert(postKey[t]);
                R.stage[t] = 0;
                R.accepted.push_back(t); accNow.push_back(t);
                open[(size_t)fpi[t]] = wave;
            }
        }
        double w2 = level_now();
        t_walk += w2 - w1;
        if (!accNow.empty()) {
            std::vector<const Patch3d*> ap(accNow.size());
            for (size_t k = 0; k < accNow.size(); k++) ap[k] = ptr[accNow[k]];
            if (!scene_p->setDepths(ap.data(), ap.size())) return false;
        }
        t_set += level_now() - w2;
        if (level_times()) std::fprintf(stderr, "extendLevel wave %d: %zu pending (%zu refined), %zu accepted, %zu deferred\n",
                                        R.waves, pending.size(), todo.size(), accNow.size(), deferred.size());
        pending.swap(deferred);
    }
    if (level_times())
        std::fprintf(stderr, "extendLevel %zu candidates (%zu refined): candidates' leaves %.1f ms, refinement %.1f ms, footprints %.1f ms, "
                             "conflict graph %.1f ms (%zu + %zu edges), waves: gates %.1f ms, walk %.1f ms, setDepths %.1f ms\n", T, rptr.size(),
                     1e3 * (t_pre - t_begin), 1e3 * (t_ref - t_pre), 1e3 * (t_fp - t_ref), 1e3 * (t_graph - t_fp), G.flow_adj.size(), G.anti_adj.size(),
                     1e3 * t_gates, 1e3 * t_walk, 1e3 * t_set);
    std::sort(R.accepted.begin(), R.accepted.end());
    return true;
}

}  // namespace mo3d
