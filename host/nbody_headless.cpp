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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/nbody_hip.h"

static int die(nbody_ctx* ctx, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, nbody_last_error(ctx));
    if (ctx) nbody_destroy(ctx);
    return 1;
}

int main(int argc, char** argv) {
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
        else { std::fprintf(stderr, "usage: %s [--mode cpu|gpu|bh] [--n N] [--frames F] [--report R] [--seed S] [--dump FILE]\n", argv[0]); return 2; }
    }
    nbody_config cfg;
    nbody_config_default(&cfg);
    cfg.n_bodies = n;
    if (mode == "cpu") { cfg.law = NBODY_LAW_CPU_R6; cfg.integrator = NBODY_INT_CPU_EULER_XBUF; }
    std::vector<float> masses(n), pos(4 * n);
    nbody_gen_refcube(seed, 0, n, cfg.world, masses.data(), pos.data());  // src/nbody_gpu.rs:61-81
    nbody_ctx* ctx = nullptr;
    int rc = nbody_create(&ctx, &cfg);
    if (rc) return die(nullptr, "nbody_create", rc);
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
            std::printf("frame %d: body0 = (%.9g, %.9g, %.9g)\n", f, pos[0], pos[1], pos[2]);
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
    std::printf("%s: n=%llu frames=%d  %.3f ms/frame  %.1f frames/s  octree nodes (last) %llu\n", mode.c_str(),
                (unsigned long long)n, frames, 1e3 * s / frames, frames / s, (unsigned long long)c.octree_nodes);
    if (out) {
        std::fseek(out, 8, SEEK_SET);
        std::fwrite(&dumped, sizeof dumped, 1, out);
        std::fclose(out);
    }
    nbody_destroy(ctx);
    return 0;
}
