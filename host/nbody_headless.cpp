// Headless host mirroring the step blocks of the reference's three binaries over the C ABI:
//   --mode cpu : src/nbody_cpu.rs:319-360   (CPU law, cross-buffer Euler)
//   --mode gpu : src/nbody_gpu.rs:418-451   (WGSL law, Verlet)
//   --mode bh  : src/nbody_gpu_bh.rs:445-527 (octree build + traversal, clamp)
// The window, camera and ray tracer of the reference are not reproduced: each "frame" is one
// call of nbody_step_*; every `--report` frames the positions are read back, as the Barnes-Hut
// binary does every frame (src/nbody_gpu_bh.rs:496-516).
// --law cpu_r6|wgsl_bias|plummer overrides the mode's force law (--mode gpu --law plummer is the
// softened Newtonian law, whose all-pairs step uses the pair kernel from N = 65,536).
// --dataset refcube|plummer: the reference's initial conditions (src/nbody_gpu.rs:61-81, seeded) or
// the Plummer sphere of the benchmarks.
// --dump FILE writes what the reference's renderer binds (src/trace.wgsl:23-27): a 16-byte header
// {"NBF1", u32 n, u32 frames written, u32 0}, masses f32[n], then per reported frame the positions
// as f32[n][4] (vec3 stride 16, pad lane 0), so that a viewer can replay the run.
// --state FILE writes the final state for comparisons between runs: {"NBS1", u32 n, u32 0, u32 0}, then
// positions, velocities and accelerations, each f32[n][4].
// --ranks R runs R processes, one per GPU (device = rank mod --devices, default R), that exchange
// inside the library (nbody_comm_init): the all-pairs modes split the targets into R index ranges,
// the Barnes-Hut mode keeps all bodies on every rank and splits the traversal (with --domains L, L = the
// depth of the cells the bodies are dealt by: every rank builds the octree of its own domain of bodies
// only, and the ranks exchange the parts of their trees the others can reach; with one rank, --domains L
// builds the tree as 8 domains on the one GPU).  Rank 0 makes the
// communicator's id and leaves it in a file for the others; rank 0 reports.  The process started
// by the user only supervises: every rank is a child forked before anything touches the GPU, and
// when one of them fails the others are ended and the id file removed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <csignal>
#include <cstring>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>
#include "../include/nbody_hip.h"

namespace {

struct Options {
    int ranks = 1, devices = 0, domains = 0;
    bool force_comm = false;
    std::string mode = "gpu", law, dataset = "refcube", dump, state;
    uint64_t n = 250;  // N_BODIES, src/nbody_gpu.rs:14
    int frames = 100, report = 50;
    uint64_t seed = 42;
};

int die(nbody_ctx* ctx, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, nbody_last_error(ctx));
    if (ctx) nbody_destroy(ctx);
    return 1;
}

// Rank 0 writes the 128-byte id to `path` (via a rename, so a reader never sees half of it); the
// others wait for the file.
bool share_id(const std::string& path, int rank, unsigned char id[128]) {
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

// One rank's whole run (the only code that touches the library's GPU entry points).
int run_rank(const Options& o, int rank, const std::string& id_path) {
    const uint64_t n = o.n;
    nbody_config cfg;
    nbody_config_default(&cfg);
    cfg.n_bodies = n;
    if (o.mode == "cpu") { cfg.law = NBODY_LAW_CPU_R6; cfg.integrator = NBODY_INT_CPU_EULER_XBUF; }
    if (o.law == "cpu_r6") cfg.law = NBODY_LAW_CPU_R6;
    else if (o.law == "wgsl_bias") cfg.law = NBODY_LAW_WGSL_BIAS;
    else if (o.law == "plummer") cfg.law = NBODY_LAW_PLUMMER;
    if (o.ranks > 1) {
        cfg.device = rank % o.devices;
        if (o.mode != "bh") {  // contiguous near-equal index ranges; Barnes-Hut keeps all bodies on every rank
            const uint64_t base = n / o.ranks, extra = n % o.ranks;
            cfg.shard_begin = rank * base + std::min<uint64_t>(rank, extra);
            cfg.shard_count = base + ((uint64_t)rank < extra ? 1 : 0);
        }
    }
    std::vector<float> masses(n), pos(4 * n);
    if (o.dataset == "plummer")  // the benchmarks' sphere: a = 12.5, cut at 8 a, the reference's expected total mass
        nbody_gen_plummer(o.seed, 0, n, n, cfg.world, 12.5f, 100.0f, 4515.625f, cfg.G, 0, masses.data(), pos.data(), nullptr);
    else
        nbody_gen_refcube(o.seed, 0, n, cfg.world, masses.data(), pos.data());  // src/nbody_gpu.rs:61-81
    nbody_ctx* ctx = nullptr;
    int rc = nbody_create(&ctx, &cfg);
    if (rc) return die(nullptr, "nbody_create", rc);
    if (o.ranks > 1 || o.force_comm) {
        unsigned char id[128];
        if (!share_id(id_path, rank, id)) { std::fprintf(stderr, "rank %d: no communicator id\n", rank); nbody_destroy(ctx); return 1; }
        if ((rc = nbody_comm_init(ctx, id, rank, o.ranks))) return die(ctx, "nbody_comm_init", rc);
    }
    if ((rc = nbody_upload(ctx, masses.data(), pos.data(), nullptr, nullptr))) return die(ctx, "nbody_upload", rc);
    // one domain per rank; without a communicator, 8 domains on the one GPU
    const uint32_t n_domains = o.domains > 0 ? ((o.ranks > 1 || o.force_comm) ? (uint32_t)o.ranks : 8u) : 0u;
    if (n_domains >= 2 && (rc = nbody_bh_set_domains(ctx, n_domains, (uint32_t)o.domains))) return die(ctx, "nbody_bh_set_domains", rc);
    std::FILE* out = nullptr;
    uint32_t dumped = 0;
    if (rank == 0 && !o.dump.empty()) {
        out = std::fopen(o.dump.c_str(), "wb");
        if (!out) { std::fprintf(stderr, "cannot open %s\n", o.dump.c_str()); nbody_destroy(ctx); return 1; }
        const uint32_t head[4] = {0x3146424eu /* "NBF1" */, (uint32_t)n, 0u, 0u};
        std::fwrite(head, sizeof head, 1, out);
        std::fwrite(masses.data(), sizeof(float), n, out);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 1; f <= o.frames; ++f) {
        rc = (o.mode == "bh") ? nbody_step_bh(ctx, 1) : nbody_step_allpairs(ctx, 1);
        if (rc) return die(ctx, "step", rc);
        if (o.report > 0 && f % o.report == 0) {
            if ((rc = nbody_download(ctx, pos.data(), nullptr, nullptr))) return die(ctx, "nbody_download", rc);
            if (rank == 0) std::printf("frame %d: body0 = (%.9g, %.9g, %.9g)\n", f, pos[0], pos[1], pos[2]);
            if (out) {
                std::fwrite(pos.data(), sizeof(float), 4 * n, out);
                ++dumped;
            }
        }
    }
    if ((rc = nbody_sync(ctx))) return die(ctx, "nbody_sync", rc);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    nbody_counters_t c;
    nbody_counters(ctx, &c);
    if (rank == 0)
        std::printf("%s: n=%llu ranks=%d frames=%d  %.3f ms/frame  %.1f frames/s  octree nodes (last) %llu  all-pairs path %llu\n",
                    o.mode.c_str(), (unsigned long long)n, o.ranks, o.frames, 1e3 * s / o.frames, o.frames / s,
                    (unsigned long long)c.octree_nodes, (unsigned long long)c.allpairs_path);
    if (n_domains >= 2 && rank == 0) {
        nbody_domain_stats_t st;
        if ((rc = nbody_bh_domain_stats(ctx, &st))) return die(ctx, "nbody_bh_domain_stats", rc);
        unsigned long long received = 0;
        for (uint32_t k = 0; k < st.domains; ++k) received += k == (uint32_t)rank ? 0 : st.exported_nodes[k];
        std::printf("domains: %u of depth-%u cells, %s; rank 0: own tree %llu nodes, received %llu nodes, top tree %llu nodes, merged %llu\n",
                    st.domains, st.top_levels, st.distributed ? "distributed (one per rank)" : "on one GPU",
                    (unsigned long long)st.nodes[0], received, (unsigned long long)st.top_nodes, (unsigned long long)st.merged_nodes);
    }
    if (out) {
        std::fseek(out, 8, SEEK_SET);
        std::fwrite(&dumped, sizeof dumped, 1, out);
        std::fclose(out);
    }
    if (rank == 0 && !o.state.empty()) {
        std::vector<float> vel(4 * n), acc(4 * n);
        if ((rc = nbody_download(ctx, pos.data(), vel.data(), acc.data()))) return die(ctx, "nbody_download", rc);
        std::FILE* f = std::fopen(o.state.c_str(), "wb");
        if (!f) { std::fprintf(stderr, "cannot open %s\n", o.state.c_str()); nbody_destroy(ctx); return 1; }
        const uint32_t head[4] = {0x3153424eu /* "NBS1" */, (uint32_t)n, 0u, 0u};
        std::fwrite(head, sizeof head, 1, f);
        std::fwrite(pos.data(), sizeof(float), 4 * n, f);
        std::fwrite(vel.data(), sizeof(float), 4 * n, f);
        std::fwrite(acc.data(), sizeof(float), 4 * n, f);
        std::fclose(f);
    }
    nbody_destroy(ctx);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return (i + 1 < argc) ? argv[++i] : (char*)"0"; };
        if (a == "--mode") o.mode = next();
        else if (a == "--law") o.law = next();
        else if (a == "--dataset") o.dataset = next();
        else if (a == "--n") o.n = std::strtoull(next(), nullptr, 10);
        else if (a == "--frames") o.frames = std::atoi(next());
        else if (a == "--report") o.report = std::atoi(next());
        else if (a == "--seed") o.seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--dump") o.dump = next();
        else if (a == "--state") o.state = next();
        else if (a == "--ranks") o.ranks = std::atoi(next());
        else if (a == "--devices") o.devices = std::atoi(next());
        else if (a == "--domains") o.domains = std::atoi(next());
        else if (a == "--comm") o.force_comm = true;  // a communicator even with one rank
        else {
            std::fprintf(stderr,
                         "usage: %s [--mode cpu|gpu|bh] [--law cpu_r6|wgsl_bias|plummer] [--dataset refcube|plummer] [--n N] "
                         "[--frames F] [--report R] [--seed S] [--dump FILE] [--state FILE] [--ranks R] [--devices D] [--comm] [--domains L]\n",
                         argv[0]);
            return 2;
        }
    }
    if (o.mode != "cpu" && o.mode != "gpu" && o.mode != "bh") { std::fprintf(stderr, "--mode must be cpu, gpu or bh\n"); return 2; }
    if (!o.law.empty() && o.law != "cpu_r6" && o.law != "wgsl_bias" && o.law != "plummer") {
        std::fprintf(stderr, "--law must be cpu_r6, wgsl_bias or plummer\n");
        return 2;
    }
    if (o.dataset != "refcube" && o.dataset != "plummer") { std::fprintf(stderr, "--dataset must be refcube or plummer\n"); return 2; }
    if (o.ranks < 1 || (uint64_t)o.ranks > o.n) { std::fprintf(stderr, "--ranks must be in [1, n]\n"); return 2; }
    if (o.domains && o.mode != "bh") { std::fprintf(stderr, "--domains goes with --mode bh\n"); return 2; }
    if (o.frames < 1) { std::fprintf(stderr, "--frames must be at least 1\n"); return 2; }
    if (o.devices <= 0) o.devices = o.ranks;
    const char* transport = std::getenv("NBODY_RCCL_LIB");
    if (o.devices < o.ranks && !(transport && *transport)) {
        // RCCL refuses two ranks on one device: such a run could only fail or hang in nbody_comm_init
        std::fprintf(stderr, "--devices %d < --ranks %d: RCCL needs one GPU per rank\n", o.devices, o.ranks);
        return 2;
    }
    const std::string id_path = "/tmp/nbody_headless_id." + std::to_string((long)getpid());
    if (o.ranks == 1) {
        const int rc = run_rank(o, 0, id_path);
        if (o.force_comm) std::remove(id_path.c_str());
        return rc;
    }
    // one process per rank, forked before anything touches the GPU; this process only supervises
    std::vector<pid_t> children;
    for (int r = 0; r < o.ranks; ++r) {
        std::fflush(nullptr);
        pid_t pid = fork();
        if (pid < 0) {
            std::perror("fork");
            for (pid_t c : children) kill(c, SIGKILL);
            for (pid_t c : children) waitpid(c, nullptr, 0);
            return 1;
        }
        if (pid == 0) {
            const int rc = run_rank(o, r, id_path);
            std::fflush(nullptr);
            _exit(rc);
        }
        children.push_back(pid);
    }
    int status = 0;
    size_t left = children.size();
    while (left > 0) {
        int st = 0;
        const pid_t done = wait(&st);
        if (done < 0) break;
        auto it = std::find(children.begin(), children.end(), done);
        if (it == children.end()) continue;
        *it = -1;
        --left;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (status == 0) {
                // a rank that failed leaves the others waiting in a collective (or for the id): end them
                std::fprintf(stderr, "a rank failed: ending the other %zu\n", left);
                for (pid_t c : children)
                    if (c > 0) kill(c, SIGKILL);
            }
            status = 1;
        }
    }
    std::remove(id_path.c_str());
    std::remove((id_path + ".tmp").c_str());
    return status;
}
