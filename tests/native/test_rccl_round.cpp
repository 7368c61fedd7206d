// One expansion round driven from a C++ host WITHOUT Python, the way north_star describes it: refine this rank's
// slice on the GPU (hpmvs_optimize_batch, device arrays), pack the refined patches into 192-byte records, ONE
// ncclAllGather over the caller's RCCL communicator, unpack.  Writes the gathered records and the unpacked arrays to a file for
// tests/test_gpu_cpp_interface.py.
//   test_rccl_round <scene dump> <out file>
// One process per GPU.  RANK / LOCAL_RANK / WORLD_SIZE come from the environment (what `python -m torch.distributed.run
// --no-python --nproc-per-node N tests/native/test_rccl_round <dump> <out>` or mpirun set; absent = one rank): rank r takes
// device LOCAL_RANK and the contiguous shard [r*n/N ...) of the round (the same cut as hpmvs_amd.distributed.shard_range);
// the communicator is built from an ncclUniqueId that rank 0 writes to $HPMVS_NCCL_ID_FILE (default /tmp/hpmvs_nccl_id.<MASTER_PORT>)
// and the others wait for.  Ragged shards travel padded to the largest shard (ncclAllGather wants equal counts); every rank
// ends with the whole round, rank 0 writes <out file>, rank r > 0 writes <out file>.rank<r> -- all of them must be equal.
// The GPU box of this repo has one GPU, so the test suite runs it with one rank; INTEGRATION.md has the 8-GPU line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <unistd.h>

#include "../../include/hpmvs_amd.h"

template <typename T> static T rd(FILE* f) { T v; if (fread(&v, sizeof(T), 1, f) != 1) { perror("read"); exit(2); } return v; }
#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed: %d %s\n", #x, rc_, hpmvs_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }

// the cut of hpmvs_amd/distributed.py: shard_range (the first n % world ranks get one patch more)
static void shard_range(int n, int rank, int world, int* lo, int* hi) {
    const int q = n / world, r = n % world;
    *lo = rank * q + (rank < r ? rank : r);
    *hi = *lo + q + (rank < r ? 1 : 0);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local_rank = env_int("LOCAL_RANK", rank);
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad RANK / WORLD_SIZE\n"); return 2; }
    HK(hipSetDevice(local_rank));
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int nv = rd<int>(f);
    hpmvs_scene* sc = nullptr;
    CK(hpmvs_scene_create(nv, local_rank, &sc));
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
    if (world == 1) {
        int dev0 = local_rank;
        if (ncclCommInitAll(&comm, 1, &dev0) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    } else {   // one process per GPU: the id travels through a file
        std::string idf = getenv("HPMVS_NCCL_ID_FILE") ? getenv("HPMVS_NCCL_ID_FILE")
                                                       : std::string("/tmp/hpmvs_nccl_id.") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
        ncclUniqueId id;
        if (rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) { fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
            const std::string tmp = idf + ".tmp";
            FILE* g = fopen(tmp.c_str(), "wb");
            if (!g || fwrite(&id, sizeof(id), 1, g) != 1) { perror("id file"); return 1; }
            fclose(g);
            if (rename(tmp.c_str(), idf.c_str()) != 0) { perror("rename"); return 1; }
        } else {
            FILE* g = nullptr;
            for (int t = 0; t < 1200 && !(g = fopen(idf.c_str(), "rb")); t++) usleep(100000);
            if (!g || fread(&id, sizeof(id), 1, g) != 1) { fprintf(stderr, "rank %d: no id file %s\n", rank, idf.c_str()); return 1; }
            fclose(g);
        }
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
        if (rank == 0) { usleep(200000); unlink(idf.c_str()); }
        int seen = 0;
        ncclCommCount(comm, &seen);
        printf("rank %d of %d on device %d: communicator of %d ranks\n", rank, world, local_rank, seen);
    }
    int lo = 0, hi = n;
    shard_range(n, rank, world, &lo, &hi);
    const int mine = hi - lo, per = (n + world - 1) / world;   // records per rank on the wire: the largest shard
    hipStream_t st;
    HK(hipStreamCreate(&st));

    // device-resident batch (what a host that keeps its frontier on the GPU hands over)
    hpmvs_patch_batch b, u;
    memset(&b, 0, sizeof(b)); memset(&u, 0, sizeof(u));
    b.n = mine; u.n = n; b.max_images = u.max_images = M;   // b: this rank's shard; u: the whole round, unpacked
    auto dalloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes); };
    HK(dalloc((void**)&b.center, 16 * n)); HK(dalloc((void**)&b.normal, 16 * n)); HK(dalloc((void**)&b.scale, 4 * n));
    HK(dalloc((void**)&b.n_images, 4 * n)); HK(dalloc((void**)&b.images, 4 * (size_t)n * M)); HK(dalloc((void**)&b.ok, n));
    HK(dalloc((void**)&b.color, 12 * n)); HK(dalloc((void**)&b.fmin, 8 * n));
    HK(dalloc((void**)&u.center, 16 * n)); HK(dalloc((void**)&u.normal, 16 * n)); HK(dalloc((void**)&u.scale, 4 * n));
    HK(dalloc((void**)&u.n_images, 4 * n)); HK(dalloc((void**)&u.images, 4 * (size_t)n * M)); HK(dalloc((void**)&u.ok, n));
    HK(dalloc((void**)&u.color, 12 * n)); HK(dalloc((void**)&u.fmin, 8 * n));
    HK(hipMemcpyAsync(b.center, center.data() + 4 * (size_t)lo, 16 * (size_t)mine, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.normal, normal.data() + 4 * (size_t)lo, 16 * (size_t)mine, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.scale, scale.data() + lo, 4 * (size_t)mine, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.n_images, nimg.data() + lo, 4 * (size_t)mine, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(b.images, images.data() + (size_t)lo * M, 4 * (size_t)mine * M, hipMemcpyHostToDevice, st));
    HK(hipMemsetAsync(b.ok, 0, n, st)); HK(hipMemsetAsync(b.color, 0, 12 * n, st)); HK(hipMemsetAsync(b.fmin, 0, 8 * n, st));
    hpmvs_options o;
    hpmvs_default_options(&o);
    hpmvs_record *send = nullptr, *recv = nullptr;
    hpmvs_record* wire = nullptr;   // what the collective delivers: `per` records from every rank (the tail of a short shard is padding)
    HK(hipMalloc((void**)&send, sizeof(hpmvs_record) * (size_t)per)); HK(hipMalloc((void**)&wire, sizeof(hpmvs_record) * (size_t)per * world));
    HK(hipMalloc((void**)&recv, sizeof(hpmvs_record) * n));
    HK(hipMemsetAsync(send, 0, sizeof(hpmvs_record) * (size_t)per, st));
    if (mine > 0) {
        CK(hpmvs_optimize_batch(sc, &o, &b, /*on_device=*/1, st));
        CK(hpmvs_pack_records(&b, send, st));
    }
    {   // refused before the collective starts: host memory as a buffer, a null communicator
        std::vector<hpmvs_record> host_buf(n);
        if (hpmvs_allgather_records(comm, host_buf.data(), (size_t)per, wire, st) != HPMVS_ERR_ARG ||
            hpmvs_allgather_records(comm, send, (size_t)per, host_buf.data(), st) != HPMVS_ERR_ARG ||
            hpmvs_allgather_records(nullptr, send, (size_t)per, wire, st) != HPMVS_ERR_ARG) {
            fprintf(stderr, "allgather_records accepted a host pointer / null communicator\n");
            return 3;
        }
    }
    CK(hpmvs_allgather_records(comm, send, (size_t)per, wire, st));   // ONE collective per round
    for (int r = 0; r < world; r++) {   // drop the padding: rank r's shard goes to its place in the round
        int rlo, rhi;
        shard_range(n, r, world, &rlo, &rhi);
        if (rhi > rlo) HK(hipMemcpyAsync(recv + rlo, wire + (size_t)r * per, sizeof(hpmvs_record) * (size_t)(rhi - rlo), hipMemcpyDeviceToDevice, st));
    }
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
    const std::string outp = rank == 0 ? std::string(argv[2]) : std::string(argv[2]) + ".rank" + std::to_string(rank);
    FILE* g = fopen(outp.c_str(), "wb");
    fwrite(&n, 4, 1, g);
    fwrite(rec.data(), sizeof(hpmvs_record), n, g);
    fwrite(un.data(), 4, n, g); fwrite(ui.data(), 4, (size_t)n * M, g); fwrite(uok.data(), 1, n, g); fwrite(uc.data(), 4, 4 * n, g);
    fclose(g);
    ncclCommDestroy(comm);
    hpmvs_scene_destroy(sc);
    printf("round ok: %d records (rank %d of %d refined %d)\n", n, rank, world, mine);
    return 0;
}
