// C ABI of libnbody_hip.so (declared in include/nbody_hip.h): context, buffers, step drivers.
#include "ctx.h"
#include <cstring>
#include <cmath>
#include <new>

thread_local std::string g_nbody_create_error;

namespace {

__global__ void k_pack_posm(const float4* __restrict__ pos16, const float* __restrict__ masses, float4* __restrict__ posm,
                            int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pos16[i];
    posm[i] = make_float4(p.x, p.y, p.z, masses[i]);
}

__global__ void k_strip_pad(const float4* __restrict__ posm, float4* __restrict__ pos16, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = posm[i];
    pos16[i] = make_float4(p.x, p.y, p.z, 0.0f);
}

// Host-side allocations (std::vector, std::string, new) inside the library must not let an
// exception cross the C boundary.
template <class F>
int guarded(nbody_ctx* ctx, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return nb_fail(ctx, NBODY_ERR_NOMEM, "out of host memory");
    } catch (...) {
        return nb_fail(ctx, NBODY_ERR_HIP, "unexpected exception inside the library");
    }
}

int set_device(nbody_ctx* ctx) {
    NB_HIP(ctx, hipSetDevice(ctx->device));
    return NBODY_OK;
}

// Positions pass through a buffer of their own (never through NBODY_BUF_ACC_PARTIAL, whose
// contents a host may be exchanging between nbody_allpairs_partial / nbody_bh_forces_ranked and
// the integrate call that follows).
int ensure_stage(nbody_ctx* ctx) {
    if (ctx->stage) return NBODY_OK;
    if (hipMalloc(&ctx->stage, ctx->n * sizeof(float4)) != hipSuccess) {
        ctx->stage = nullptr;
        (void)hipGetLastError();
        return nb_fail(ctx, NBODY_ERR_NOMEM, "position staging buffer: out of device memory");
    }
    return NBODY_OK;
}

int upload_set(nbody_ctx* ctx, int s, const float* masses, const float* pos16, const float* vel16, const float* acc16,
               bool need_pos) {
    const size_t b16 = ctx->n * sizeof(float4);
    if (int r = ensure_stage(ctx)) return r;
    if (pos16) {
        // pos16 -> device, then pack the mass into the pad lane
        NB_HIP(ctx, hipMemcpyAsync(ctx->stage, pos16, b16, hipMemcpyHostToDevice, ctx->stream));
        if (masses) NB_HIP(ctx, hipMemcpyAsync(ctx->masses, masses, ctx->n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        unsigned nb = (unsigned)((ctx->n + 255) / 256);
        hipLaunchKernelGGL(k_pack_posm, dim3(nb), dim3(256), 0, ctx->stream, ctx->stage, ctx->masses, ctx->posm[s],
                           (int)ctx->n);
        NB_HIP(ctx, hipGetLastError());
    } else if (need_pos) {
        return nb_fail(ctx, NBODY_ERR_INVALID, "positions are required");
    } else {
        // zero positions, masses kept in the pad lane
        NB_HIP(ctx, hipMemsetAsync(ctx->stage, 0, b16, ctx->stream));
        unsigned nb = (unsigned)((ctx->n + 255) / 256);
        hipLaunchKernelGGL(k_pack_posm, dim3(nb), dim3(256), 0, ctx->stream, ctx->stage, ctx->masses, ctx->posm[s],
                           (int)ctx->n);
        NB_HIP(ctx, hipGetLastError());
    }
    if (vel16) NB_HIP(ctx, hipMemcpyAsync(ctx->vel[s], vel16, b16, hipMemcpyHostToDevice, ctx->stream));
    else NB_HIP(ctx, hipMemsetAsync(ctx->vel[s], 0, b16, ctx->stream));
    if (acc16) NB_HIP(ctx, hipMemcpyAsync(ctx->acc[s], acc16, b16, hipMemcpyHostToDevice, ctx->stream));
    else NB_HIP(ctx, hipMemsetAsync(ctx->acc[s], 0, b16, ctx->stream));
    // the host buffers may be pageable: make sure the copies have consumed them
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NBODY_OK;
}

int download_set(nbody_ctx* ctx, int s, float* pos16, float* vel16, float* acc16) {
    const size_t b16 = ctx->n * sizeof(float4);
    if (pos16) {
        if (int r = ensure_stage(ctx)) return r;
        unsigned nb = (unsigned)((ctx->n + 255) / 256);
        hipLaunchKernelGGL(k_strip_pad, dim3(nb), dim3(256), 0, ctx->stream, ctx->posm[s], ctx->stage, (int)ctx->n);
        NB_HIP(ctx, hipGetLastError());
        NB_HIP(ctx, hipMemcpyAsync(pos16, ctx->stage, b16, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (vel16) NB_HIP(ctx, hipMemcpyAsync(vel16, ctx->vel[s], b16, hipMemcpyDeviceToHost, ctx->stream));
    if (acc16) NB_HIP(ctx, hipMemcpyAsync(acc16, ctx->acc[s], b16, hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NBODY_OK;
}

}  // namespace

extern "C" {

const char* nbody_version(void) { return "nbody_hip 0.1 (gfx950)"; }

const char* nbody_last_error(const nbody_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    // calls without a context: nbody_create, or nbody_accel_allpairs_noalloc if that failed since
    const char* m = nb_noalloc_last_error();
    return *m ? m : g_nbody_create_error.c_str();
}

void nbody_config_default(nbody_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->n_bodies = 250;  // N_BODIES, src/nbody_gpu.rs:14
    cfg->law = NBODY_LAW_WGSL_BIAS;
    cfg->integrator = NBODY_INT_WGSL_VERLET;
    cfg->G = 0.0066743f;
    cfg->dt = 0.1f;
    cfg->eps2 = 1.0f;
    cfg->theta = 0.5f;
    cfg->world = 250.0f;
    cfg->max_depth = 16;
    cfg->leaflist_mode = 0;
    cfg->shard_begin = 0;
    cfg->shard_count = 0;
    cfg->device = -1;
    cfg->stream = nullptr;
}

int nbody_create(nbody_ctx** out, const nbody_config* cfg) {
    if (!out || !cfg) {
        g_nbody_create_error = "nbody_create: null argument";
        return NBODY_ERR_INVALID;
    }
    *out = nullptr;
    nb_noalloc_clear_error();
    if (cfg->n_bodies == 0 || cfg->n_bodies > (1ull << 30)) {
        g_nbody_create_error = "nbody_create: n_bodies must be in [1, 2^30]";
        return NBODY_ERR_INVALID;
    }
    if (cfg->law < 0 || cfg->law > 2 || cfg->integrator < 0 || cfg->integrator > 2) {
        g_nbody_create_error = "nbody_create: unknown law or integrator";
        return NBODY_ERR_INVALID;
    }
    if (cfg->quadrupole != 0 && (cfg->quadrupole != 1 || cfg->law != NBODY_LAW_PLUMMER)) {
        g_nbody_create_error = "nbody_create: quadrupole is 0 or 1, and 1 needs the PLUMMER law";
        return NBODY_ERR_INVALID;
    }
    if (cfg->max_depth > 16) {
        g_nbody_create_error = "nbody_create: max_depth must be in [1,16], or 0 for the uncapped src/octree.rs variant";
        return NBODY_ERR_INVALID;
    }
    uint64_t sc = cfg->shard_count ? cfg->shard_count : cfg->n_bodies - cfg->shard_begin;
    if (cfg->shard_begin >= cfg->n_bodies || sc > cfg->n_bodies - cfg->shard_begin) {
        g_nbody_create_error = "nbody_create: shard range outside [0, n_bodies)";
        return NBODY_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_nbody_create_error = "nbody_create: no HIP device";
        return NBODY_ERR_NO_DEVICE;
    }
    nbody_ctx* ctx = new (std::nothrow) nbody_ctx;
    if (!ctx) {
        g_nbody_create_error = "nbody_create: out of host memory";
        return NBODY_ERR_NOMEM;
    }
    ctx->cfg = *cfg;
    ctx->n = cfg->n_bodies;
    ctx->i_begin = cfg->shard_begin;
    ctx->i_count = sc;
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    ctx->device = dev;
    auto fail = [&](int code, const std::string& msg) {
        g_nbody_create_error = msg;
        nbody_destroy(ctx);
        return code;
    };
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess) return fail(NBODY_ERR_NO_DEVICE, "nbody_create: bad device ordinal");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(NBODY_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NBODY_ERR_NO_DEVICE, std::string("nbody_create: device is ") + prop.gcnArchName + ", this library is built for gfx950");
    ctx->num_cus = prop.multiProcessorCount;
    if (cfg->stream) {
        ctx->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            return fail(NBODY_ERR_HIP, "hipStreamCreate failed");
        ctx->own_stream = true;
    }
    const size_t b16 = ctx->n * sizeof(float4);
    hipError_t e = hipSuccess;
    for (int s = 0; s < 2 && e == hipSuccess; ++s) {
        e = hipMalloc(&ctx->posm[s], b16);
        if (e == hipSuccess) e = hipMalloc(&ctx->vel[s], b16);
        if (e == hipSuccess) e = hipMalloc(&ctx->acc[s], b16);
    }
    if (e == hipSuccess) e = hipMalloc(&ctx->acc_tmp, b16);
    if (e == hipSuccess) e = hipMalloc(&ctx->masses, ctx->n * sizeof(float));
    if (e != hipSuccess) return fail(NBODY_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    for (int s = 0; s < 2; ++s) {
        hipMemsetAsync(ctx->posm[s], 0, b16, ctx->stream);
        hipMemsetAsync(ctx->vel[s], 0, b16, ctx->stream);
        hipMemsetAsync(ctx->acc[s], 0, b16, ctx->stream);
    }
    hipMemsetAsync(ctx->masses, 0, ctx->n * sizeof(float), ctx->stream);
    if (guarded(ctx, [&]() -> int { return nb_tree_create(ctx); }) != NBODY_OK)
        return fail(NBODY_ERR_NOMEM, "octree workspace: " + ctx->err);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(NBODY_ERR_HIP, "stream sync failed");
    *out = ctx;
    return NBODY_OK;
}

void nbody_destroy(nbody_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (int which = 0; which < 3; ++which) nb_time_collect(ctx, which);
    nb_comm_destroy(ctx);
    nb_tree_destroy(ctx);
    nb_sym_destroy(ctx);
    for (int s = 0; s < 2; ++s) {
        if (ctx->posm[s]) hipFree(ctx->posm[s]);
        if (ctx->vel[s]) hipFree(ctx->vel[s]);
        if (ctx->acc[s]) hipFree(ctx->acc[s]);
    }
    if (ctx->acc_tmp) hipFree(ctx->acc_tmp);
    if (ctx->stage) hipFree(ctx->stage);
    if (ctx->acc_part) hipFree(ctx->acc_part);
    if (ctx->acc_body) hipFree(ctx->acc_body);
    if (ctx->masses) hipFree(ctx->masses);
    if (ctx->energy_dev) hipFree(ctx->energy_dev);
    if (ctx->clk_dev) hipFree(ctx->clk_dev);
    for (auto& v : ctx->ev)
        for (hipEvent_t e : v) hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int nbody_upload(nbody_ctx* ctx, const float* host_masses, const float* host_pos16, const float* host_vel16,
                 const float* host_acc16) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!host_masses || !host_pos16) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload: masses and positions are required");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->cur = 0;
        ctx->leapfrog_started = false;
        ctx->pending_forces = 0;
        nb_tree_forget_order(ctx);
        ctx->mass0 = host_masses[0];
        ctx->uniform_mass = true;
        for (uint64_t i = 1; i < ctx->n && ctx->uniform_mass; ++i) ctx->uniform_mass = host_masses[i] == ctx->mass0;
        if (ctx->cfg.max_depth == 0) {
            // src/octree.rs:49 takes a node of total mass 0 for an EMPTY one: the next body inserted there
            // replaces a massless body instead of splitting the cell.  That is not reproduced; refused instead.
            for (uint64_t i = 0; i < ctx->n; ++i)
                if (host_masses[i] == 0.0f)
                    return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload: max_depth 0 (the octree.rs builder) does not take bodies of mass 0");
        }
        r = upload_set(ctx, 0, host_masses, host_pos16, host_vel16, host_acc16, true);
        if (r) return r;
        // second set: zeros (src/nbody_cpu.rs:58-60), mass kept in the pad lane
        r = upload_set(ctx, 1, nullptr, nullptr, nullptr, nullptr, false);
        if (r) return r;
        ctx->uploaded = true;
        return NBODY_OK;
    });
}

int nbody_upload_alt(nbody_ctx* ctx, const float* host_pos16, const float* host_vel16, const float* host_acc16) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_upload_alt: call nbody_upload first");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return upload_set(ctx, ctx->cur ^ 1, nullptr, host_pos16, host_vel16, host_acc16, false);
    });
}

namespace {

const char* bh_refusal(const nbody_ctx* ctx) {
    if (ctx->cfg.law == NBODY_LAW_CPU_R6) return "the Barnes-Hut path uses the WGSL laws (src/nbodybh.wgsl), not CPU_R6";
    if (ctx->cfg.integrator == NBODY_INT_CPU_EULER_XBUF)
        return "the Barnes-Hut path integrates as src/nbodybh.wgsl:163-169 (or leapfrog), not with the cross-buffer Euler update";
    return nullptr;
}

// The pair kernel dealt over the ranks applies where it does on one GPU: a symmetric law and
// enough bodies; every rank evaluates the same condition.
bool pair_sharded(const nbody_ctx* ctx) {
    return ctx->cfg.law == NBODY_LAW_PLUMMER && ctx->cfg.integrator != NBODY_INT_CPU_EULER_XBUF && ctx->n >= 65536;
}

int ranked_integrate(nbody_ctx* ctx) {
    if (!ctx->acc_body && hipMalloc(&ctx->acc_body, ctx->n * sizeof(float4)) != hipSuccess) {
        ctx->acc_body = nullptr;
        (void)hipGetLastError();
        return nb_fail(ctx, NBODY_ERR_NOMEM, "accelerations in body order: out of device memory");
    }
    int r = nb_tree_unrank(ctx, ctx->acc_tmp, ctx->acc_body);
    if (r) return r;
    r = nb_integrate_clamped(ctx, ctx->acc_body);
    if (r) return r;
    ctx->counters.steps_bh += 1;
    return NBODY_OK;
}

}  // namespace

int nbody_step_allpairs(nbody_ctx* ctx, int nsteps) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_step_allpairs: no bodies uploaded");
    if (nsteps < 0) return nb_fail(ctx, NBODY_ERR_INVALID, "nsteps < 0");
    if (ctx->i_count < ctx->n && nsteps > 1 && !ctx->comm)
        return nb_fail(ctx, NBODY_ERR_INVALID,
                       "a sharded context without a communicator steps once per call: all-gather NBODY_BUF_POSM between steps");
    return guarded(ctx, [&]() -> int {
        int rc = set_device(ctx);
        if (rc) return rc;
        ctx->pending_forces = 0;
        // The pair kernel dealt over the ranks only if EVERY rank has room for its scratch: a rank
        // that took the other path alone would leave the rest waiting in the all-reduce.  (Any
        // other per-rank failure between two collectives does exactly that: it ends the run.)
        bool pair = false;
        if (ctx->comm && pair_sharded(ctx) && nsteps > 0) {
            rc = nb_comm_pair_path(ctx, nb_allpairs_pair_fits(ctx), &pair);
            if (rc) return rc;
        }
        for (int s = 0; s < nsteps; ++s) {
            if (pair) {
                rc = nb_allpairs_partial(ctx, (uint32_t)nb_comm_rank(ctx), (uint32_t)nb_comm_world(ctx));
                if (rc) return rc;
                rc = nb_comm_allreduce_partial(ctx);
            } else {
                rc = nb_allpairs_forces(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp);
            }
            if (rc) return rc;
            rc = nb_integrate(ctx, ctx->acc_tmp);
            if (rc) return rc;
            if (ctx->comm) {
                rc = nb_comm_gather_positions(ctx);
                if (rc) return rc;
            }
            ctx->counters.steps_allpairs += 1;
        }
        return NBODY_OK;
    });
}

int nbody_allpairs_partial(nbody_ctx* ctx, uint32_t part, uint32_t nparts) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_allpairs_partial: no bodies uploaded");
    if (ctx->cfg.law != NBODY_LAW_PLUMMER || ctx->cfg.integrator == NBODY_INT_CPU_EULER_XBUF)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_allpairs_partial: only the PLUMMER law (symmetric in the pair), not the cross-buffer integrator");
    if (nparts == 0 || part >= nparts) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_allpairs_partial: part >= nparts");
    if (ctx->n < 4097) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_allpairs_partial: needs at least two superblocks of bodies (N > 4096)");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        r = nb_allpairs_partial(ctx, part, nparts);
        if (r) return r;
        ctx->pending_forces = 1;
        return NBODY_OK;
    });
}

int nbody_integrate_partial(nbody_ctx* ctx) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_integrate_partial: no bodies uploaded");
    if (ctx->pending_forces != 1)
        return nb_fail(ctx, NBODY_ERR_STATE,
                       "nbody_integrate_partial: NBODY_BUF_ACC_PARTIAL does not hold the forces of nbody_allpairs_partial (another call has used the buffer since)");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        r = nb_integrate(ctx, ctx->acc_tmp);
        if (r) return r;
        ctx->counters.steps_allpairs += 1;
        return NBODY_OK;
    });
}

int nbody_step_bh(nbody_ctx* ctx, int nsteps) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_step_bh: no bodies uploaded");
    if (nsteps < 0) return nb_fail(ctx, NBODY_ERR_INVALID, "nsteps < 0");
    if (const char* why = bh_refusal(ctx)) return nb_fail(ctx, NBODY_ERR_INVALID, why);
    if (ctx->comm && ctx->i_count < ctx->n)
        return nb_fail(ctx, NBODY_ERR_INVALID,
                       "nbody_step_bh with a communicator needs a context without a shard range (every rank holds all bodies)");
    if (ctx->i_count < ctx->n && nsteps > 1)
        return nb_fail(ctx, NBODY_ERR_INVALID, "a sharded context steps once per call: all-gather NBODY_BUF_POSM between steps");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        for (int s = 0; s < nsteps; ++s) {
            if (ctx->comm && nb_tree_distributed(ctx)) {
                // domains: this rank builds its own domain's tree, essential trees and accelerations are exchanged
                if (!ctx->acc_body && hipMalloc(&ctx->acc_body, ctx->n * sizeof(float4)) != hipSuccess) {
                    ctx->acc_body = nullptr;
                    (void)hipGetLastError();
                    return nb_fail(ctx, NBODY_ERR_NOMEM, "accelerations in body order: out of device memory");
                }
                r = nb_tree_forces_distributed(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_body);
                if (r) return r;
                r = nb_integrate_clamped(ctx, ctx->acc_body);
                if (r) return r;
                ctx->counters.steps_bh += 1;
                continue;
            }
            r = nb_tree_build(ctx, ctx->posm[ctx->cur]);
            if (r) return r;
            if (ctx->comm) {
                // this rank's run of the sorted order, the runs gathered by rank, all bodies integrated
                uint64_t rb = 0, rc = 0;
                nb_comm_run(ctx, nb_comm_rank(ctx), &rb, &rc);
                r = nb_tree_walk_ranked(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp, rb, rc);
                if (r) return r;
                r = nb_comm_gather_ranked(ctx);
                if (r) return r;
                r = ranked_integrate(ctx);
                if (r) return r;
                continue;
            }
            r = nb_tree_walk(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp);
            if (r) return r;
            r = nb_integrate_clamped(ctx, ctx->acc_tmp);
            if (r) return r;
            ctx->counters.steps_bh += 1;
        }
        return NBODY_OK;
    });
}

int nbody_bh_forces_ranked(nbody_ctx* ctx, uint64_t rank_begin, uint64_t rank_count) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_bh_forces_ranked: no bodies uploaded");
    if (const char* why = bh_refusal(ctx)) return nb_fail(ctx, NBODY_ERR_INVALID, why);
    if (ctx->i_count != ctx->n)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_forces_ranked: the context must hold all bodies (no shard range)");
    if (rank_begin > ctx->n || rank_count > ctx->n - rank_begin)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_forces_ranked: rank range outside the bodies");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        r = nb_tree_build(ctx, ctx->posm[ctx->cur]);
        if (r) return r;
        r = nb_tree_walk_ranked(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp, rank_begin, rank_count);
        if (r) return r;
        ctx->pending_forces = 2;
        return NBODY_OK;
    });
}

int nbody_integrate_ranked(nbody_ctx* ctx) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_integrate_ranked: no bodies uploaded");
    if (ctx->i_count != ctx->n)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_integrate_ranked: the context must hold all bodies (no shard range)");
    if (const char* why = bh_refusal(ctx)) return nb_fail(ctx, NBODY_ERR_INVALID, why);
    if (ctx->pending_forces != 2)
        return nb_fail(ctx, NBODY_ERR_STATE,
                       "nbody_integrate_ranked: NBODY_BUF_ACC_PARTIAL does not hold the forces of nbody_bh_forces_ranked (another call has used the buffer since)");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        return ranked_integrate(ctx);
    });
}

int nbody_accel_allpairs(nbody_ctx* ctx, float* host_acc16_out) {
    if (!ctx || !host_acc16_out) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_accel_allpairs: no bodies uploaded");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        r = nb_allpairs_forces(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp);
        if (r) return r;
        NB_HIP(ctx, hipMemcpyAsync(host_acc16_out + 4 * ctx->i_begin, ctx->acc_tmp + ctx->i_begin,
                                   ctx->i_count * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return NBODY_OK;
    });
}

int nbody_accel_bh(nbody_ctx* ctx, float* host_acc16_out) {
    if (!ctx || !host_acc16_out) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_accel_bh: no bodies uploaded");
    if (ctx->cfg.law == NBODY_LAW_CPU_R6)
        return nb_fail(ctx, NBODY_ERR_INVALID, "the Barnes-Hut path uses the WGSL laws (src/nbodybh.wgsl), not CPU_R6");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        ctx->pending_forces = 0;
        r = nb_tree_walk(ctx, ctx->posm[ctx->cur], ctx->vel[ctx->cur], ctx->acc_tmp);
        if (r) return r;
        NB_HIP(ctx, hipMemcpyAsync(host_acc16_out + 4 * ctx->i_begin, ctx->acc_tmp + ctx->i_begin,
                                   ctx->i_count * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return nb_tree_check(ctx);
    });
}

int nbody_build_octree(nbody_ctx* ctx) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_build_octree: no bodies uploaded");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_build(ctx, ctx->posm[ctx->cur]);
    });
}

int nbody_download_octree(nbody_ctx* ctx, nbody_octree_node* host_nodes, uint64_t cap_nodes, uint64_t* n_nodes) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_download(ctx, host_nodes, cap_nodes, n_nodes);
    });
}

int nbody_upload_octree(nbody_ctx* ctx, const nbody_octree_node* host_nodes, uint64_t n_nodes) {
    if (!ctx || !host_nodes || n_nodes == 0) return ctx ? nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: empty tree") : NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_upload(ctx, host_nodes, n_nodes);
    });
}

int nbody_morton_keys(nbody_ctx* ctx, uint64_t* host_keys, uint8_t* host_leaf_depth, uint32_t* host_order) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_keys(ctx, host_keys, host_leaf_depth, host_order);
    });
}

int nbody_bh_set_target_order(nbody_ctx* ctx, const uint32_t* host_order, uint64_t count) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (ctx->i_count != ctx->n) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_target_order: the context must hold all bodies (no shard range)");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_set_target_order(ctx, host_order, count);
    });
}

int nbody_bh_set_domains(nbody_ctx* ctx, uint32_t domains, uint32_t top_levels) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_set_domains(ctx, domains, top_levels);
    });
}

int nbody_bh_set_domain_view(nbody_ctx* ctx, int32_t domain) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_set_domain_view(ctx, (int)domain);
    });
}

int nbody_bh_domain_stats(nbody_ctx* ctx, nbody_domain_stats_t* out) {
    if (!ctx || !out) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_tree_domain_stats(ctx, out);
    });
}

int nbody_download(nbody_ctx* ctx, float* host_pos16, float* host_vel16, float* host_acc16) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_download: no bodies uploaded");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        r = download_set(ctx, ctx->cur, host_pos16, host_vel16, host_acc16);
        if (r) return r;
        return nb_tree_check(ctx);
    });
}

int nbody_download_alt(nbody_ctx* ctx, float* host_pos16, float* host_vel16, float* host_acc16) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_download_alt: no bodies uploaded");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return download_set(ctx, ctx->cur ^ 1, host_pos16, host_vel16, host_acc16);
    });
}

int nbody_sync(nbody_ctx* ctx) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return nb_tree_check(ctx);
    });
}

int nbody_energy(nbody_ctx* ctx, double* kinetic, double* potential) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (!ctx->uploaded) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_energy: no bodies uploaded");
    if (ctx->i_count < ctx->n) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_energy needs an unsharded context");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_energy(ctx, kinetic, potential);
    });
}

int nbody_counters(nbody_ctx* ctx, nbody_counters_t* out) {
    if (!ctx || !out) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *out = ctx->counters;
        return NBODY_OK;
    });
}

int nbody_counters_reset(nbody_ctx* ctx) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        ctx->counters = nbody_counters_t{};
        return NBODY_OK;
    });
}

int nbody_set_counting(nbody_ctx* ctx, int on) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        ctx->counting = on != 0;
        return NBODY_OK;
    });
}

int nbody_device_buffer(nbody_ctx* ctx, int which, void** dev_ptr, size_t* bytes) {
    if (!ctx || !dev_ptr) return NBODY_ERR_INVALID;
    size_t b16 = ctx->n * sizeof(float4);
    switch (which) {
        case NBODY_BUF_POSM: *dev_ptr = ctx->posm[ctx->cur]; if (bytes) *bytes = b16; return NBODY_OK;
        case NBODY_BUF_VEL: *dev_ptr = ctx->vel[ctx->cur]; if (bytes) *bytes = b16; return NBODY_OK;
        case NBODY_BUF_ACC: *dev_ptr = ctx->acc[ctx->cur]; if (bytes) *bytes = b16; return NBODY_OK;
        case NBODY_BUF_OCTREE:
            // (the export allocates and launches: on the context's device)
            return guarded(ctx, [&]() -> int {
                int r = set_device(ctx);
                if (r) return r;
                return nb_tree_device_nodes(ctx, dev_ptr, bytes);
            });
        case NBODY_BUF_ACC_PARTIAL: *dev_ptr = ctx->acc_tmp; if (bytes) *bytes = b16; return NBODY_OK;
        default: return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_device_buffer: unknown buffer");
    }
}

int nbody_read_buffer(nbody_ctx* ctx, int which, void* host_dst, size_t bytes) {
    if (!ctx || !host_dst) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        void* dev = nullptr;
        size_t have = 0;
        r = nbody_device_buffer(ctx, which, &dev, &have);
        if (r) return r;
        if (bytes > have) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_read_buffer: more bytes requested than the buffer holds");
        NB_HIP(ctx, hipMemcpyAsync(host_dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return NBODY_OK;
    });
}

void* nbody_stream(nbody_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int nbody_set_kernel_timing(nbody_ctx* ctx, int on) {
    if (!ctx) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        if (on) {
            // everything a timed launch needs exists before the first one: a pool of event pairs and
            // the clock stamp words of the pair kernel
            r = nb_time_reserve(ctx, 64);
            if (r) return r;
            if (!ctx->clk_dev) {
                NB_HIP(ctx, hipMalloc(&ctx->clk_dev, 2 * sizeof(unsigned long long)));
                NB_HIP(ctx, hipMemsetAsync(ctx->clk_dev, 0, 2 * sizeof(unsigned long long), ctx->stream));
            }
        }
        ctx->timing = on != 0;
        return NBODY_OK;
    });
}

int nbody_kernel_clock(nbody_ctx* ctx, double* mhz) {
    if (!ctx || !mhz) return NBODY_ERR_INVALID;
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        *mhz = 0.0;
        if (!ctx->clk_dev) return NBODY_OK;
        unsigned long long h[2] = {0, 0};
        NB_HIP(ctx, hipMemcpyAsync(h, ctx->clk_dev, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h[1] > 0) *mhz = 100.0 * (double)h[0] / (double)h[1];
        return NBODY_OK;
    });
}

int nbody_kernel_time(nbody_ctx* ctx, int which, double* avg_ms, uint64_t* launches) {
    if (!ctx) return NBODY_ERR_INVALID;
    if (which < 0 || which > 2) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_kernel_time: which must be 0, 1 or 2");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        r = nb_time_collect(ctx, which);
        if (r) return r;
        if (launches) *launches = ctx->time_launches[which];
        if (avg_ms) *avg_ms = ctx->time_launches[which] ? ctx->time_acc_ms[which] / (double)ctx->time_launches[which] : 0.0;
        ctx->time_acc_ms[which] = 0;
        ctx->time_launches[which] = 0;
        return NBODY_OK;
    });
}

int nbody_comm_unique_id(void* id_out128) {
    if (!id_out128) return NBODY_ERR_INVALID;
    nb_noalloc_clear_error();  // nbody_last_error(NULL) reports the last call made without a context
    try {
        int r = nb_comm_unique_id(id_out128);
        if (r) g_nbody_create_error = "nbody_comm_unique_id: librccl could not be loaded or ncclGetUniqueId failed";
        return r;
    } catch (...) {
        return NBODY_ERR_NOMEM;
    }
}

int nbody_comm_sorted_run(uint64_t n_bodies, int world, int rank, uint64_t* begin, uint64_t* count) {
    if (!begin || !count || world < 1 || rank < 0 || rank >= world) return NBODY_ERR_INVALID;
    nb_sorted_run(n_bodies, (uint64_t)world, (uint64_t)rank, begin, count);
    return NBODY_OK;
}

int nbody_comm_exchange_plan(uint64_t n_bodies, int world, const uint64_t* begin, const uint64_t* count) {
    if (!begin || !count || world < 1) return NBODY_ERR_INVALID;
    const int plan = nb_exchange_plan(n_bodies, world, begin, count);
    return plan == NB_PLAN_INVALID ? NBODY_ERR_INVALID : plan;
}

int nbody_comm_init(nbody_ctx* ctx, const void* unique_id128, int rank, int world) {
    if (!ctx || !unique_id128) return NBODY_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_comm_init: rank outside [0, world)");
    return guarded(ctx, [&]() -> int {
        int r = set_device(ctx);
        if (r) return r;
        return nb_comm_init(ctx, unique_id128, rank, world);
    });
}

}  // extern "C"
