// One expansion round driven from a C++ host WITHOUT Python, the way north_star describes it: refine this rank's
// slice on the GPU (hpmvs_optimize_batch, device arrays), pack the refined patches into 192-byte records, ONE
// ncclAllGather over the caller's RCCL communicator, unpack.  Run here with a 1-rank communicator (the GPU box has
// one GPU); writes the gathered records and the unpacked arrays to a file for tests/test_gpu_cpp_interface.py.
//   test_rccl_round <scene dump> <out file>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hpmvs_amd.h"

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) { perror("read"); exit(2); } return v; }
#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed: %d %s\n", #x, rc_, hpmvs_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int nv = rd<int>(f);
    hpmvs_scene* sc = nullptr;
    CK(hpmvs_scene_create(nv, 0, &sc));
    for (int i = 0; i < nv; i++) {
        const int W = rd<int>(f), H = rd<int>(f);
        const double fl = rd<double>(f);
        double q[4], c[3];
        for (int k = 0; k < 4; k++) q[k] = rd<double>(f);
        for (int k = 0; k < 3; k++) c[k] = rd<double>(f);
        std::vector<unsigned char> pix((size_t)W * H * 3);
        if (fread(pix.data(), 1, pix.size(), f) != pix.size()) return 2;
        hpmvs_camera cam;
        CK(hpmvs_camera_from_nvm(fl, q, c, W, H, 5, &cam));
        CK(hpmvs_scene_set_view(sc, i, W, H, pix.data(), 0, &cam));
    }
    for (int i = 0; i < nv; i++) {
        const int m = rd<int>(f);
        std::vector<int32_t> ids(m > 0 ? m : 1);
        for (int k = 0; k < m; k++) ids[k] = rd<int>(f);
        CK(hpmvs_scene_set_covis(sc, i, ids.data(), m));
    }
    CK(hpmvs_scene_commit(sc));
    const int n = rd<int>(f), M = 32;
    std::vector<float> center(4 * n), normal(4 * n), scale(n);
    std::vector<int32_t> nimg(n), images((size_t)n * M, -1);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 4; k++) center[4 * i + k] = rd<float>(f);
        for (int k = 0; k < 4; k++) normal[4 * i + k] = rd<float>(f);
        scale[i] = rd<float>(f);
        nimg[i] = rd<int>(f);
        for (int k = 0; k < nimg[i]; k++) images[(size_t)i * M + k] = rd<int>(f);
    }
    fclose(f);

    ncclComm_t comm;
    int dev0 = 0;
    if (ncclCommInitAll(&comm, 1, &dev0) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    hipStream_t st;
    HK(hipStreamCreate(&st));

    // device-resident batch (what a host that keeps its frontier on the GPU hands over)
    hpmvs_patch_batch b, u;
    memset(&b, 0, sizeof(b)); memset(&u, 0, sizeof(u));
    b.n = u.n = n; b.max_images = u.max_images = M;
    auto dalloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes); };
    HK(dalloc((void**)&b.center, 16 * n)); HK(dalloc((void**)&b.normal, 16 * n)); HK(dalloc((void**)&b.scale, 4 * n));
    HK(dalloc((void**)&b.n_images, 4 * n)); HK(dalloc((void**)&b.images, 4 * (size_t)n * M)); HK(dalloc((void**)&b.ok, n));
    HK(dalloc((void**)&b.color, 12 * n)); HK(dalloc((void**)&b.fmin, 8 * n));
    HK(dalloc((void**)&u.center, 16 * n)); HK(dalloc((void**)&u.normal, 16 * n)); HK(dalloc((void**)&u.scale, 4 * n));
    HK(dalloc((void**)&u.n_images, 4 * n)); HK(dalloc((void**)&u.images, 4 * (size_t)n * M)); HK(dalloc((void**)&u.ok, n));
    HK(dalloc((void**)&u.color, 12 * n)); HK(dalloc((void**)&u.fmin, 8 * n));
    HK(hipMemcpyAsync(b.center, center.data(), 16 * n, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.normal, normal.data(), 16 * n, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.scale, scale.data(), 4 * n, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.n_images, nimg.data(), 4 * n, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.images, images.data(), 4 * (size_t)n * M, hipMemcpyHostToDevice, st));
    HK(hipMemsetAsync(b.ok, 0, n, st)); HK(hipMemsetAsync(b.color, 0, 12 * n, st)); HK(hipMemsetAsync(b.fmin, 0, 8 * n, st));
    hpmvs_options o;
    hpmvs_default_options(&o);
    hpmvs_record *send = nullptr, *recv = nullptr;
    HK(hipMalloc((void**)&send, sizeof(hpmvs_record) * n)); HK(hipMalloc((void**)&recv, sizeof(hpmvs_record) * n));
    CK(hpmvs_optimize_batch(sc, &o, &b, /*on_device=*/1, st));
    CK(hpmvs_pack_records(&b, send, st));
    {   // refused before the collective starts: host memory as a buffer, a null communicator
        std::vector<hpmvs_record> host_buf(n);
        if (hpmvs_allgather_records(comm, host_buf.data(), (size_t)n, recv, st) != HPMVS_ERR_ARG ||
            hpmvs_allgather_records(comm, send, (size_t)n, host_buf.data(), st) != HPMVS_ERR_ARG ||
            hpmvs_allgather_records(nullptr, send, (size_t)n, recv, st) != HPMVS_ERR_ARG) {
            fprintf(stderr, "allgather_records accepted a host pointer / null communicator\n");
            return 3;
        }
    }
    CK(hpmvs_allgather_records(comm, send, (size_t)n, recv, st));
    CK(hpmvs_unpack_records(recv, n, &u, st));
    HK(hipStreamSynchronize(st));

    std::vector<hpmvs_record> rec(n);
    HK(hipMemcpy(rec.data(), recv, sizeof(hpmvs_record) * n, hipMemcpyDeviceToHost));
    std::vector<int32_t> un(n), ui((size_t)n * M);
    std::vector<unsigned char> uok(n);
    std::vector<float> uc(4 * n);
    HK(hipMemcpy(un.data(), u.n_images, 4 * n, hipMemcpyDeviceToHost));
    HK(hipMemcpy(ui.data(), u.images, 4 * (size_t)n * M, hipMemcpyDeviceToHost));
    HK(hipMemcpy(uok.data(), u.ok, n, hipMemcpyDeviceToHost));
    HK(hipMemcpy(uc.data(), u.center, 16 * n, hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb");
    fwrite(&n, 4, 1, g);
    fwrite(rec.data(), sizeof(hpmvs_record), n, g);
    fwrite(un.data(), 4, n, g); fwrite(ui.data(), 4, (size_t)n * M, g); fwrite(uok.data(), 1, n, g); fwrite(uc.data(), 4, 4 * n, g);
    fclose(g);
    ncclCommDestroy(comm);
    hpmvs_scene_destroy(sc);
    printf("round ok: %d records\n", n);
    return 0;
}
