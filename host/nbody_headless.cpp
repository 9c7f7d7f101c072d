// Headless host mirroring the step blocks of the reference's three binaries over the C ABI:
//   --mode cpu : src/nbody_cpu.rs:319-360   (CPU law, cross-buffer Euler)
//   --mode gpu : src/nbody_gpu.rs:418-451   (WGSL law, Verlet)
//   --mode bh  : src/nbody_gpu_bh.rs:445-527 (octree build + traversal, clamp)
// The window, camera and ray tracer of the reference are not reproduced: each "frame" is one
// call of nbody_step_*; every `--report` frames the positions are read back, as the Barnes-Hut
// binary does every frame (src/nbody_gpu_bh.rs:496-516).
// --dump FILE writes what the reference's renderer binds (src/trace.wgsl:23-27): a 16-byte header
// {"NBF1", u32 n, u32 frames written, u32 0}, masses f32[n], then per reported frame the positions
// as f32[n][4] (vec3 stride 16, pad lane 0), so that a viewer can replay the run.
// --ranks R runs R processes, one per GPU (device = rank mod device count), that exchange inside the
// library over RCCL (nbody_comm_init): the all-pairs modes split the targets into R index ranges,
// the Barnes-Hut mode keeps all bodies on every rank and splits the traversal.  Rank 0 makes the
// communicator's id and leaves it in a file for the others; rank 0 reports.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>
#include "../include/nbody_hip.h"

static int die(nbody_ctx* ctx, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, nbody_last_error(ctx));
    if (ctx) nbody_destroy(ctx);
    return 1;
}

// Rank 0 writes the 128-byte id to `path` (via a rename, so a reader never sees half of it); the
// others wait for the file.
static bool share_id(const std::string& path, int rank, unsigned char id[128]) {
    if (rank == 0) {
        if (nbody_comm_unique_id(id)) return false;
        const std::string tmp = path + ".tmp";
        std::FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(id, 1, 128, f) == 128;
        std::fclose(f);
        return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
    }
    for (int tries = 0; tries < 600; ++tries) {
        if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
            const bool ok = std::fread(id, 1, 128, f) == 128;
            std::fclose(f);
            if (ok) return true;
        }
        usleep(100000);
    }
    return false;
}

int main(int argc, char** argv) {
    int ranks = 1;
    bool force_comm = false;
    std::string mode = "gpu";
    uint64_t n = 250;  // N_BODIES, src/nbody_gpu.rs:14
    int frames = 100, report = 50;
    uint64_t seed = 42;
    std::string dump;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return (i + 1 < argc) ? argv[++i] : (char*)"0"; };
        if (a == "--mode") mode = next();
        else if (a == "--n") n = std::strtoull(next(), nullptr, 10);
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--report") report = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--dump") dump = next();
        else if (a == "--ranks") ranks = std::atoi(next());
        else if (a == "--comm") force_comm = true;  // a communicator even with one rank
        else { std::fprintf(stderr, "usage: %s [--mode cpu|gpu|bh] [--n N] [--frames F] [--report R] [--seed S] [--dump FILE] [--ranks R] [--comm]\n", argv[0]); return 2; }
    }
    // one process per rank, forked before anything touches the GPU
    int rank = 0;
    std::vector<pid_t> children;
    const std::string id_path = "/tmp/nbody_headless_id." + std::to_string((long)getpid());
    if (ranks < 1 || (uint64_t)ranks > n) { std::fprintf(stderr, "--ranks must be in [1, n]\n"); return 2; }
    for (int r = 1; r < ranks; ++r) {
        pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) { rank = r; children.clear(); break; }
        children.push_back(pid);
    }
    nbody_config cfg;
    nbody_config_default(&cfg);
    cfg.n_bodies = n;
    if (mode == "cpu") { cfg.law = NBODY_LAW_CPU_R6; cfg.integrator = NBODY_INT_CPU_EULER_XBUF; }
    if (ranks > 1) {
        cfg.device = rank;  // taken modulo the device count below
        if (mode != "bh") {  // contiguous near-equal index ranges; Barnes-Hut keeps all bodies on every rank
            const uint64_t base = n / ranks, extra = n % ranks;
            cfg.shard_begin = rank * base + std::min<uint64_t>(rank, extra);
            cfg.shard_count = base + ((uint64_t)rank < extra ? 1 : 0);
        }
    }
    std::vector<float> masses(n), pos(4 * n);
    nbody_gen_refcube(seed, 0, n, cfg.world, masses.data(), pos.data());  // src/nbody_gpu.rs:61-81
    nbody_ctx* ctx = nullptr;
    int rc = nbody_create(&ctx, &cfg);
    if (rc == NBODY_ERR_NO_DEVICE && ranks > 1 && cfg.device > 0) {  // fewer GPUs than ranks: share them
        cfg.device = 0;
        rc = nbody_create(&ctx, &cfg);
    }
    if (rc) return die(nullptr, "nbody_create", rc);
    if (ranks > 1 || force_comm) {
        unsigned char id[128];
        if (!share_id(id_path, rank, id)) { std::fprintf(stderr, "rank %d: no communicator id\n", rank); nbody_destroy(ctx); return 1; }
        if ((rc = nbody_comm_init(ctx, id, rank, ranks))) return die(ctx, "nbody_comm_init", rc);
    }
    if (rank != 0) dump.clear();
    if ((rc = nbody_upload(ctx, masses.data(), pos.data(), nullptr, nullptr))) return die(ctx, "nbody_upload", rc);
    std::FILE* out = nullptr;
    uint32_t dumped = 0;
    if (!dump.empty()) {
        out = std::fopen(dump.c_str(), "wb");
        if (!out) { std::fprintf(stderr, "cannot open %s\n", dump.c_str()); nbody_destroy(ctx); return 1; }
        const uint32_t head[4] = {0x3146424eu /* "NBF1" */, (uint32_t)n, 0u, 0u};
        std::fwrite(head, sizeof head, 1, out);
        std::fwrite(masses.data(), sizeof(float), n, out);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 1; f <= frames; ++f) {
        rc = (mode == "bh") ? nbody_step_bh(ctx, 1) : nbody_step_allpairs(ctx, 1);
        if (rc) return die(ctx, "step", rc);
        if (report > 0 && f % report == 0) {
            if ((rc = nbody_download(ctx, pos.data(), nullptr, nullptr))) return die(ctx, "nbody_download", rc);
            if (rank == 0) std::printf("frame %d: body0 = (%.9g, %.9g, %.9g)\n", f, pos[0], pos[1], pos[2]);
            if (out) {
                std::fwrite(pos.data(), sizeof(float), 4 * n, out);
                ++dumped;
            }
        }
    }
    nbody_sync(ctx);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    nbody_counters_t c;
    nbody_counters(ctx, &c);
    if (rank == 0)
        std::printf("%s: n=%llu ranks=%d frames=%d  %.3f ms/frame  %.1f frames/s  octree nodes (last) %llu\n", mode.c_str(),
                    (unsigned long long)n, ranks, frames, 1e3 * s / frames, frames / s, (unsigned long long)c.octree_nodes);
    if (out) {
        std::fseek(out, 8, SEEK_SET);
        std::fwrite(&dumped, sizeof dumped, 1, out);
        std::fclose(out);
    }
    nbody_destroy(ctx);
    int status = 0;
    for (pid_t pid : children) {
        int st = 0;
        waitpid(pid, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = 1;
    }
    if (rank == 0 && (ranks > 1 || force_comm)) std::remove(id_path.c_str());
    return status;
}
