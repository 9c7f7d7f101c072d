// placeholder until the octree build/walk kernels land
#include "ctx.h"
struct Octree { int dummy; };
int nb_tree_create(nbody_ctx*) { return NBODY_OK; }
void nb_tree_destroy(nbody_ctx*) {}
int nb_tree_build(nbody_ctx* ctx, const float4*) { return nb_fail(ctx, NBODY_ERR_STATE, "octree build not available"); }
int nb_tree_walk(nbody_ctx* ctx, const float4*, const float4*, float4*) { return nb_fail(ctx, NBODY_ERR_STATE, "octree walk not available"); }
int nb_tree_download(nbody_ctx* ctx, nbody_octree_node*, uint64_t, uint64_t*) { return nb_fail(ctx, NBODY_ERR_STATE, "no octree"); }
int nb_tree_upload(nbody_ctx* ctx, const nbody_octree_node*, uint64_t) { return nb_fail(ctx, NBODY_ERR_STATE, "no octree"); }
int nb_tree_keys(nbody_ctx* ctx, uint64_t*, uint8_t*, uint32_t*) { return nb_fail(ctx, NBODY_ERR_STATE, "no octree"); }
int nb_tree_device_nodes(nbody_ctx* ctx, void**, size_t*) { return nb_fail(ctx, NBODY_ERR_STATE, "no octree"); }
