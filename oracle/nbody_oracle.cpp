// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the
// product library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// A restatement, in scalar f32 C++ with the reference's evaluation order, of the gravity step of
// JMS55/nbody.  Every function cites the reference file:line it follows.  Compile with
// -ffp-contract=off and without fast-math: Rust never fuses mul+add and sums sequentially.
//
// PARITY PINNING: the reference ships no tests, fixtures or golden vectors for this path, and it
// cannot be compiled here (Rust; no cargo/rustc).  glam 0.22.0 (Cargo.lock) supplies the Vec3
// arithmetic on the CPU path; its scalar Vec3 is componentwise IEEE f32, dot = x*x+y*y+z*z
// left-to-right, max_element = x.max(y.max(z)), Sum folds left from ZERO.  This oracle is pinned
// only by hand-derived known-answer tests (tests/test_oracle_kat.py) => "parity unpinned".
//
// The WGSL builtins pow/acos/distance are implementation-defined in accuracy; here they are the
// correctly-rounded-ish libm f32 functions.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// glam Vec3::dot: (x*x) + (y*y) + (z*z), left to right.
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline V3 vmin(V3 a, V3 b) { return v3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
inline float max_element(V3 a) { return std::fmax(a.x, std::fmax(a.y, a.z)); }
// WGSL distance(a,b) = length(a-b) = sqrt(dot(d,d)).
inline float distance(V3 a, V3 b) {
    V3 d = a - b;
    return std::sqrt(dot(d, d));
}

inline V3 ld16(const float* p, size_t i) { return v3(p[4 * i], p[4 * i + 1], p[4 * i + 2]); }
inline void st16(float* p, size_t i, V3 v) {
    p[4 * i] = v.x;
    p[4 * i + 1] = v.y;
    p[4 * i + 2] = v.z;
}

// ---- force laws -------------------------------------------------------------------------------
enum Law { LAW_CPU_R6 = 0, LAW_WGSL_BIAS = 1, LAW_PLUMMER = 2 };

const float G_F32 = 0.0066743f;

// src/nbody_cpu.rs:337-340 -- one source term of the CPU law.
//   distance = p2 - p; distance_norm = distance.dot(distance);
//   (0.0066743 * m2 * distance) / (dn * dn * dn)
inline V3 term_cpu_r6(V3 p, V3 p2, float m2) {
    V3 d = p2 - p;
    float dn = dot(d, d);
    return ((G_F32 * m2) * d) / (dn * dn * dn);
}

// src/nbody.wgsl:54-81 (= src/nbodybh.wgsl:77-102) -- one source term of the WGSL law.
inline V3 term_wgsl(V3 pos, V3 vel, V3 other_pos, float other_mass, bool with_bias, float G, float eps2) {
    V3 dist_vec = other_pos - pos;
    float dist = distance(other_pos, pos);
    float divisor = std::pow(dist, 2.0f);  // :59
    divisor += eps2;                       // :60
    divisor = std::pow(divisor, 1.5f);     // :62
    float g = G * other_mass / divisor;    // :65
    if (with_bias) {
        // :74-78  acos( dot(vel,d) / (|d|*|vel| + 1) ) ^ 0.15
        float bias = std::acos(dot(vel, dist_vec) /
                               (distance(v3(0, 0, 0), dist_vec) * distance(v3(0, 0, 0), vel) + 1.0f));
        bias = std::pow(bias, 0.15f);
        g *= bias;  // :79
    }
    return g * dist_vec;  // :81
}

// Same term evaluated in f64 from the f32 inputs: used only to measure rounding error.
inline void term_f64(int law, const double p[3], const double v[3], const double q[3], double m, double G,
                     double eps2, double out[3]) {
    double d[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
    double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double s;
    if (law == LAW_CPU_R6) {
        s = G * m / (r2 * r2 * r2);
    } else {
        s = G * m / std::pow(r2 + eps2, 1.5);
        if (law == LAW_WGSL_BIAS) {
            double vd = v[0] * d[0] + v[1] * d[1] + v[2] * d[2];
            double vn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            double b = std::acos(vd / (std::sqrt(r2) * vn + 1.0));
            s *= std::pow(b, 0.15);
        }
    }
    out[0] = s * d[0];
    out[1] = s * d[1];
    out[2] = s * d[2];
}

// ---- octree -----------------------------------------------------------------------------------
// src/octree_maxdepth.rs:13-23 with the WGSL storage layout encase gives it
// (src/nbodybh.wgsl:1-9): offsets 0,16,32,44,48,52..84,84; 96 bytes.
// src/octree.rs:5-15 shares the layout with is_leaf at 84 and max_depth at 88.
struct Node96 {
    float com[3];
    float _p0;
    float pos_min[3];
    float _p1;
    float pos_max[3];
    float range;
    float total_mass;
    uint32_t child[8];
    uint32_t node_type;  // octree.rs: is_leaf
    uint32_t max_depth;  // octree.rs only; padding for octree_maxdepth.rs
    uint32_t _p2;
};
static_assert(sizeof(Node96) == 96, "node layout");

const uint32_t T_DUMMY = 0, T_LEAFBODY = 1, T_LEAFLIST = 2, T_INTERIOR = 3;

inline V3 gcom(const Node96& n) { return v3(n.com[0], n.com[1], n.com[2]); }
inline void set3(float* d, V3 v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}

// src/octree_maxdepth.rs:210-222
inline unsigned node_index_for_child(V3 c, V3 p) {
    unsigned i = 0;
    if (p.x >= c.x) i |= 4;
    if (p.y >= c.y) i |= 2;
    if (p.z >= c.z) i |= 1;
    return i;
}
// src/octree_maxdepth.rs:224-230
inline V3 extent_weights(unsigned ci) {
    return v3(-1.0f + (2.0f * (float)((ci & 4) >> 2)), -1.0f + (2.0f * (float)((ci & 2) >> 1)),
              -1.0f + (2.0f * (float)(ci & 1)));
}

struct Builder {
    std::vector<Node96> nodes;
    // leaf lists: per list, the owning node index and the buffered children
    std::vector<uint32_t> list_owner;
    std::vector<std::vector<Node96>> list_children;
    uint32_t max_depth_cap;  // 16 for octree_maxdepth.rs
    int leaflist_mode;       // 0 = literal reference behaviour, 1 = intended (every body stored)
    float world;
    bool overflow = false;

    static Node96 dummy() {
        Node96 n;
        std::memset(&n, 0, sizeof n);
        return n;
    }
    // src/octree_maxdepth.rs:188-194
    void ensure_has_child(uint32_t self, unsigned ci) {
        if (nodes[self].child[ci] == 0) {
            uint32_t i = (uint32_t)nodes.size();
            nodes.push_back(dummy());
            nodes[self].child[ci] = i;
        }
    }
    // src/octree_maxdepth.rs:75-86 applied to a free-standing node
    static void make_leafbody(Node96& n, V3 p, float m) {
        n.total_mass = m;
        set3(n.com, p);
        set3(n.pos_min, p);
        set3(n.pos_max, p);
        n.range = max_element(v3(n.pos_max[0], n.pos_max[1], n.pos_max[2]) -
                              v3(n.pos_min[0], n.pos_min[1], n.pos_min[2]));
        n.node_type = T_LEAFBODY;
    }

    // src/octree_maxdepth.rs:63-186.  `self` is an index, not a reference: this restates the
    // intended semantics (as if `nodes` never reallocated) rather than the dangling-&mut UB.
    void insert(uint32_t self, V3 position, float mass, V3 self_center, float self_extents, uint32_t depth) {
        if (nodes[self].node_type == T_DUMMY) {  // :75-86
            make_leafbody(nodes[self], position, mass);
            return;
        }
        // :91-101
        V3 a_pos = gcom(nodes[self]);
        float a_mass = nodes[self].total_mass;
        {
            Node96& n = nodes[self];
            n.total_mass = a_mass + mass;
            set3(n.com, ((a_pos * a_mass) + (position * mass)) / n.total_mass);
            set3(n.pos_min, vmin(v3(n.pos_min[0], n.pos_min[1], n.pos_min[2]), position));
            set3(n.pos_max, vmax(v3(n.pos_max[0], n.pos_max[1], n.pos_max[2]), position));
            n.range = max_element(v3(n.pos_max[0], n.pos_max[1], n.pos_max[2]) -
                                  v3(n.pos_min[0], n.pos_min[1], n.pos_min[2]));
        }
        if (depth == max_depth_cap) {  // :103-136
            Node96& n = nodes[self];
            bool fresh = n.node_type != T_LEAFLIST;
            if (fresh) {  // :114-120
                n.node_type = T_LEAFLIST;
                n.child[0] = (uint32_t)list_children.size();
                n.child[1] = 0;
                list_children.emplace_back();
                list_owner.push_back(self);
            }
            auto& lst = list_children[n.child[0]];
            if (leaflist_mode == 0) {
                // :123-136 literal: the child is built from the node's PREVIOUS aggregate
                // (node_a_*), not from the new body.
                Node96 c = dummy();
                make_leafbody(c, a_pos, a_mass);
                lst.push_back(c);
                n.child[1] += 1;
            } else {
                // intended: every body of the cell is a list member
                if (fresh) {
                    Node96 c = dummy();
                    make_leafbody(c, a_pos, a_mass);
                    lst.push_back(c);
                    n.child[1] += 1;
                }
                Node96 c = dummy();
                make_leafbody(c, position, mass);
                lst.push_back(c);
                n.child[1] += 1;
            }
            return;
        }
        if (depth >= 200) {  // unbounded variant only: coincident bodies never separate
            overflow = true;
            return;
        }
        // :139-184
        float half = self_extents / 2.0f;
        unsigned ci_b = node_index_for_child(self_center, position);
        ensure_has_child(self, ci_b);
        if (nodes[self].node_type == T_LEAFBODY) {
            nodes[self].node_type = T_INTERIOR;
            unsigned ci_a = node_index_for_child(self_center, a_pos);
            ensure_has_child(self, ci_a);
            insert(nodes[self].child[ci_a], a_pos, a_mass, self_center + (half * extent_weights(ci_a)), half,
                   depth + 1);
        }
        insert(nodes[self].child[ci_b], position, mass, self_center + (half * extent_weights(ci_b)), half,
               depth + 1);
    }

    // Leaf of a body in the finished tree: descend by the same octant rule until a non-interior
    // node (LEAFBODY or LEAFLIST) is reached.
    void locate(V3 p, uint32_t* node_out, uint8_t* depth_out) const {
        float ext = world / 2.0f;
        V3 c = v3(ext, ext, ext);
        uint32_t idx = 0;
        uint32_t depth = 0;
        while (nodes[idx].node_type == T_INTERIOR) {
            ext = ext / 2.0f;
            unsigned ci = node_index_for_child(c, p);
            c = c + (ext * extent_weights(ci));
            idx = nodes[idx].child[ci];
            ++depth;
        }
        *node_out = idx;
        *depth_out = (uint8_t)depth;
    }
};

}  // namespace

extern "C" {

int orc_node_size() { return (int)sizeof(Node96); }

// ---------------------------------------------------------------------------------------------
// Accelerations, all-pairs.  pos16/vel16/acc16: 16-byte stride (x,y,z,pad).
// law 0: src/nbody_cpu.rs:332-342 (self skipped, ascending n2, fold from ZERO).
// law 1: src/nbody.wgsl:50-116 (self included, ascending i).
// law 2: law 1 without the bias factor (:74-79 removed).
// accum64 != 0: each f32 term is widened and summed in f64 (isolates summation-order error).
// i_begin/i_count: the target range (all sources are always used).
void orc_accel_allpairs(int law, int64_t n, const float* pos16, const float* vel16, const float* masses,
                        float G, float eps2, int64_t i_begin, int64_t i_count, int accum64, float* acc16_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = i_begin; i < i_begin + i_count; ++i) {
        V3 p = ld16(pos16, i);
        V3 v = vel16 ? ld16(vel16, i) : v3(0, 0, 0);
        if (!accum64) {
            V3 a = v3(0, 0, 0);
            for (int64_t j = 0; j < n; ++j) {
                if (law == LAW_CPU_R6) {
                    if (j == i) continue;
                    a = a + term_cpu_r6(p, ld16(pos16, j), masses[j]);
                } else {
                    V3 t = term_wgsl(p, v, ld16(pos16, j), masses[j], law == LAW_WGSL_BIAS, G, eps2);
                    a = a + t;  // acc += g * dist_vec
                }
            }
            st16(acc16_out, i, a);
        } else {
            double a[3] = {0, 0, 0};
            for (int64_t j = 0; j < n; ++j) {
                V3 t;
                if (law == LAW_CPU_R6) {
                    if (j == i) continue;
                    t = term_cpu_r6(p, ld16(pos16, j), masses[j]);
                } else {
                    t = term_wgsl(p, v, ld16(pos16, j), masses[j], law == LAW_WGSL_BIAS, G, eps2);
                }
                a[0] += t.x;
                a[1] += t.y;
                a[2] += t.z;
            }
            st16(acc16_out, i, v3((float)a[0], (float)a[1], (float)a[2]));
        }
    }
}

// Fully-f64 accelerations from the same f32 inputs (error yardstick). acc_out: 3 doubles per body.
void orc_accel_allpairs_f64(int law, int64_t n, const float* pos16, const float* vel16, const float* masses,
                            double G, double eps2, int64_t i_begin, int64_t i_count, double* acc_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = i_begin; i < i_begin + i_count; ++i) {
        double p[3] = {pos16[4 * i], pos16[4 * i + 1], pos16[4 * i + 2]};
        double v[3] = {0, 0, 0};
        if (vel16) {
            v[0] = vel16[4 * i];
            v[1] = vel16[4 * i + 1];
            v[2] = vel16[4 * i + 2];
        }
        double a[3] = {0, 0, 0};
        for (int64_t j = 0; j < n; ++j) {
            if (law == LAW_CPU_R6 && j == i) continue;
            double q[3] = {pos16[4 * j], pos16[4 * j + 1], pos16[4 * j + 2]};
            double t[3];
            term_f64(law, p, v, q, masses[j], G, eps2, t);
            a[0] += t[0];
            a[1] += t[1];
            a[2] += t[2];
        }
        acc_out[3 * (i - i_begin)] = a[0];
        acc_out[3 * (i - i_begin) + 1] = a[1];
        acc_out[3 * (i - i_begin) + 2] = a[2];
    }
}

// Sum over sources of the NORM of each term (f64): the scale of the rounding error any f32
// evaluation of the sum carries (error <= c * eps_f32 * this), whatever the summation order.
void orc_accel_allpairs_abs(int law, int64_t n, const float* pos16, const float* vel16, const float* masses,
                            double G, double eps2, int64_t i_begin, int64_t i_count, double* abs_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = i_begin; i < i_begin + i_count; ++i) {
        double p[3] = {pos16[4 * i], pos16[4 * i + 1], pos16[4 * i + 2]};
        double v[3] = {0, 0, 0};
        if (vel16) {
            v[0] = vel16[4 * i];
            v[1] = vel16[4 * i + 1];
            v[2] = vel16[4 * i + 2];
        }
        double a = 0;
        for (int64_t j = 0; j < n; ++j) {
            if (j == i) continue;
            double q[3] = {pos16[4 * j], pos16[4 * j + 1], pos16[4 * j + 2]};
            double t[3];
            term_f64(law, p, v, q, masses[j], G, eps2, t);
            a += std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        }
        abs_out[i - i_begin] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// src/nbody_cpu.rs:321-360 -- the CPU binary's step, literally: forces from set 1, Euler update
// of set 2 (which is NOT derived from set 1: two interleaved states), then swap.
// State: p1,v1,a1,p2,v2,a2 (16-byte stride), modified in place; after the call set "1" is the
// one the next step reads (the swap is applied to the caller's arrays by copying).
void orc_step_cpu(int64_t n, int nsteps, const float* masses, float* p1, float* v1, float* a1, float* p2,
                  float* v2, float* a2) {
    std::vector<float> tmp;
    float *P1 = p1, *V1 = v1, *A1 = a1, *P2 = p2, *V2 = v2, *A2 = a2;
    for (int s = 0; s < nsteps; ++s) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            V3 p = ld16(P1, i);
            V3 a = v3(0, 0, 0);  // Sum::sum folds from Vec3::ZERO
            for (int64_t j = 0; j < n; ++j) {
                if (j == i) continue;  // :335
                a = a + term_cpu_r6(p, ld16(P1, j), masses[j]);
            }
            st16(A2, i, a);             // :345
            V3 v = ld16(V2, i) + a;     // :346
            st16(V2, i, v);             // :347
            st16(P2, i, ld16(P2, i) + v);  // :348
        }
        std::swap(P1, P2);  // :358-360
        std::swap(V1, V2);
        std::swap(A1, A2);
    }
    if (nsteps & 1) {  // make the caller's "1" arrays hold the set the next step would read
        size_t bytes = (size_t)n * 16;
        tmp.resize((size_t)n * 4);
        float* pairs[3][2] = {{p1, p2}, {v1, v2}, {a1, a2}};
        for (auto& pr : pairs) {
            std::memcpy(tmp.data(), pr[0], bytes);
            std::memcpy(pr[0], pr[1], bytes);
            std::memcpy(pr[1], tmp.data(), bytes);
        }
    }
}

// src/nbody.wgsl:132-138 (clamp_world <= 0) and src/nbodybh.wgsl:163-169 (clamp to [0,world]).
void orc_integrate_wgsl(int64_t n, float dt, float clamp_world, const float* pos_in, const float* vel_in,
                        const float* acc_in, const float* acc_new, float* pos_out, float* vel_out,
                        float* acc_out) {
    for (int64_t i = 0; i < n; ++i) {
        V3 pos = ld16(pos_in, i), vel = ld16(vel_in, i), ain = ld16(acc_in, i), acc = ld16(acc_new, i);
        // pos += vel*dt + 0.5*acc_in*pow(dt,2)
        pos = pos + ((vel * dt) + ((0.5f * ain) * std::pow(dt, 2.0f)));
        // vel += 0.5*(acc_in+acc)*dt
        vel = vel + ((0.5f * (ain + acc)) * dt);
        if (clamp_world > 0.0f) {
            pos = v3(std::fmin(std::fmax(pos.x, 0.0f), clamp_world), std::fmin(std::fmax(pos.y, 0.0f), clamp_world),
                     std::fmin(std::fmax(pos.z, 0.0f), clamp_world));
        }
        st16(pos_out, i, pos);
        st16(vel_out, i, vel);
        st16(acc_out, i, acc);
    }
}

// src/nbody_gpu.rs:418-451 + src/nbody.wgsl -- nsteps of the GPU all-pairs binary.
// State arrays are updated in place (the ping-pong is internal).
void orc_step_wgsl_allpairs(int law, int64_t n, int nsteps, const float* masses, float G, float eps2, float dt,
                            float* pos, float* vel, float* acc) {
    std::vector<float> an((size_t)n * 4, 0.f), po((size_t)n * 4, 0.f), vo((size_t)n * 4, 0.f), ao((size_t)n * 4, 0.f);
    for (int s = 0; s < nsteps; ++s) {
        orc_accel_allpairs(law, n, pos, vel, masses, G, eps2, 0, n, 0, an.data());
        orc_integrate_wgsl(n, dt, 0.0f, pos, vel, acc, an.data(), po.data(), vo.data(), ao.data());
        std::memcpy(pos, po.data(), (size_t)n * 16);
        std::memcpy(vel, vo.data(), (size_t)n * 16);
        std::memcpy(acc, ao.data(), (size_t)n * 16);
    }
}

// ---------------------------------------------------------------------------------------------
// Octree build.  variant 0: src/octree_maxdepth.rs:26-61 (depth cap 16, leaf lists);
// variant 1: src/octree.rs:18-39 (no cap; is_leaf/max_depth fields).
// pos16: 16-byte stride.  nodes_out: capacity cap_nodes (96 B each).  body_leaf/body_depth: per
// body, the node index and depth of its leaf (LEAFBODY node, or the LEAFLIST node that holds it)
// in the finished tree.  Returns the node count, or -1 if cap_nodes is too small, -2 if the
// unbounded variant did not terminate (coincident bodies), -3 if variant 1 is given a body of mass 0:
// src/octree.rs:49 recognises an empty node by `total_mass == 0.0`, so a massless body is REPLACED by the
// next body that reaches its node; this restatement tells empty nodes by their type and does not
// reproduce that.  *max_depth_out: octree.rs:36 value.
int64_t orc_build_octree(int variant, int64_t n, const float* pos16, const float* masses, float world,
                         uint32_t max_depth, int leaflist_mode, Node96* nodes_out, int64_t cap_nodes,
                         uint32_t* body_leaf, uint8_t* body_depth, uint32_t* max_depth_out) {
    if (variant == 1)
        for (int64_t i = 0; i < n; ++i)
            if (masses[i] == 0.0f) return -3;
    Builder b;
    b.max_depth_cap = (variant == 0) ? max_depth : 0xffffffffu;
    b.leaflist_mode = leaflist_mode;
    b.world = world;
    std::vector<uint32_t> leaf((size_t)n, 0);
    std::vector<uint8_t> dep((size_t)n, 0);
    b.nodes.push_back(Builder::dummy());  // :27,:32
    float root_extents = world / 2.0f;    // :28
    V3 root_center = v3(root_extents, root_extents, root_extents);
    for (int64_t i = 0; i < n; ++i) {  // :39-48 sequential, index order
        b.insert(0, ld16(pos16, i), masses[i], root_center, root_extents, 0);
        if (b.overflow) return -2;
    }
    for (int64_t i = 0; i < n; ++i) b.locate(ld16(pos16, i), &leaf[i], &dep[i]);

    // :51-56 leaf-list children appended contiguously, in list-creation order
    for (size_t li = 0; li < b.list_children.size(); ++li) {
        Node96& owner = b.nodes[b.list_owner[li]];
        owner.child[0] = (uint32_t)b.nodes.size();
        owner.child[1] = (uint32_t)b.list_children[li].size();
        for (auto& c : b.list_children[li]) b.nodes.push_back(c);
    }
    int64_t cnt = (int64_t)b.nodes.size();
    if (variant == 1) {
        // src/octree.rs: is_leaf instead of node_type; max_depth = max over inserts of
        // ceil(log2(ceil(world/ext))) of each created leaf (:58), stored at nodes[0] (:36).
        uint32_t md = 0;
        for (int64_t i = 0; i < n; ++i) {
            float ext = root_extents;
            for (uint8_t d = 0; d < dep[i]; ++d) ext = ext / 2.0f;
            uint32_t v = (uint32_t)std::ceil(std::log2(std::ceil(world / ext)));
            md = std::max(md, v);
        }
        for (auto& nd : b.nodes) {
            nd.node_type = (nd.node_type == T_LEAFBODY) ? 1u : 0u;  // is_leaf
            nd.max_depth = 0;
        }
        b.nodes[0].max_depth = md;
        if (max_depth_out) *max_depth_out = md;
    }
    if (body_leaf) std::memcpy(body_leaf, leaf.data(), (size_t)n * 4);
    if (body_depth) std::memcpy(body_depth, dep.data(), (size_t)n);
    if (cnt > cap_nodes) return -1;
    std::memcpy(nodes_out, b.nodes.data(), (size_t)cnt * sizeof(Node96));
    return cnt;
}

// 48-bit Morton key of a position by the reference's comparison descent
// (src/octree_maxdepth.rs:139,142,156 + :210-230): 16 levels of `>=` against the cell centre,
// octant bits x->4, y->2, z->1, most significant level first.
uint64_t orc_morton_key(float x, float y, float z, float world, uint32_t levels) {
    V3 p = v3(x, y, z);
    float ext = world / 2.0f;
    V3 c = v3(ext, ext, ext);
    uint64_t key = 0;
    for (uint32_t l = 0; l < levels; ++l) {
        ext = ext / 2.0f;
        unsigned ci = node_index_for_child(c, p);
        key = (key << 3) | ci;
        c = c + (ext * extent_weights(ci));
    }
    return key;
}
void orc_morton_keys(int64_t n, const float* pos16, float world, uint32_t levels, uint64_t* keys) {
    for (int64_t i = 0; i < n; ++i) keys[i] = orc_morton_key(pos16[4 * i], pos16[4 * i + 1], pos16[4 * i + 2], world, levels);
}

// ---------------------------------------------------------------------------------------------
// src/nbodybh.wgsl:63-161 -- the traversal, per body, literally (private LIFO stack, children
// pushed ascending => popped highest index first).  nodes: octree_maxdepth.rs layout.
// counters (optional, 4 x int64): node visits, leaf interactions, aggregate interactions, max stack.
// Returns 0, or -1 if a body's stack exceeded `stack_cap` entries (the reference's is 800).
int orc_bh_accel(int law, int64_t n, const float* pos16, const float* vel16, const Node96* nodes, float G,
                 float eps2, float theta, int64_t i_begin, int64_t i_count, int64_t stack_cap, float* acc16_out,
                 int64_t* counters) {
    int64_t visits = 0, leaf_int = 0, agg_int = 0, max_stack = 0;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : visits, leaf_int, agg_int) reduction(max : max_stack)
    for (int64_t i = i_begin; i < i_begin + i_count; ++i) {
        V3 pos = ld16(pos16, i);
        V3 vel = vel16 ? ld16(vel16, i) : v3(0, 0, 0);
        V3 acc = v3(0, 0, 0);
        std::vector<uint32_t> stack;
        stack.reserve(256);
        stack.push_back(0);
        bool with_bias = law == LAW_WGSL_BIAS;
        while (!stack.empty()) {
            uint32_t idx = stack.back();
            stack.pop_back();
            const Node96& node = nodes[idx];
            ++visits;
            if (node.node_type == T_LEAFBODY) {  // :76-102
                acc = acc + term_wgsl(pos, vel, gcom(node), node.total_mass, with_bias, G, eps2);
                ++leaf_int;
            } else {
                float d = node.range;                  // :104
                float r = distance(pos, gcom(node));   // :105
                if ((d / r) < theta) {                 // :106
                    acc = acc + term_wgsl(pos, vel, gcom(node), node.total_mass, with_bias, G, eps2);
                    ++agg_int;
                } else if (node.node_type == T_LEAFLIST) {  // :137-146 do-while: pushes >= 1
                    uint32_t j = node.child[0], k = 0;
                    do {
                        stack.push_back(j + k);
                        ++k;
                    } while (k < node.child[1]);
                } else {  // :148-157
                    for (unsigned k = 0; k < 8; ++k)
                        if (node.child[k] != 0u) stack.push_back(node.child[k]);
                }
                if ((int64_t)stack.size() > max_stack) max_stack = (int64_t)stack.size();
                if ((int64_t)stack.size() > stack_cap) {
#pragma omp atomic write
                    err = -1;
                    break;
                }
            }
        }
        st16(acc16_out, i, acc);
    }
    if (counters) {
        counters[0] = visits;
        counters[1] = leaf_int;
        counters[2] = agg_int;
        counters[3] = max_stack;
    }
    return err;
}

// The same traversal (identical f32 acceptance decisions, so the identical node sequence) with
// every accepted term evaluated and summed in f64: the yardstick for the walk's rounding error.
// abs_out (optional, one double per body): the sum of the accepted terms' magnitudes, the scale of the
// rounding error ANY f32 summation order of those terms carries.
void orc_bh_accel_f64(int law, int64_t n, const float* pos16, const float* vel16, const Node96* nodes, double G,
                      double eps2, float theta, int64_t i_begin, int64_t i_count, double* acc_out, double* abs_out) {
    (void)n;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = i_begin; i < i_begin + i_count; ++i) {
        V3 pos = ld16(pos16, i);
        double p[3] = {pos.x, pos.y, pos.z};
        double v[3] = {0, 0, 0};
        if (vel16) {
            v[0] = vel16[4 * i];
            v[1] = vel16[4 * i + 1];
            v[2] = vel16[4 * i + 2];
        }
        double a[3] = {0, 0, 0};
        double mag = 0;
        std::vector<uint32_t> stack;
        stack.push_back(0);
        while (!stack.empty()) {
            uint32_t idx = stack.back();
            stack.pop_back();
            const Node96& node = nodes[idx];
            bool take = node.node_type == T_LEAFBODY;
            if (!take) {
                float d = node.range;
                float r = distance(pos, gcom(node));
                take = (d / r) < theta;
            }
            if (take) {
                double q[3] = {node.com[0], node.com[1], node.com[2]};
                double t[3];
                term_f64(law, p, v, q, node.total_mass, G, eps2, t);
                a[0] += t[0];
                a[1] += t[1];
                a[2] += t[2];
                mag += std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
            } else if (node.node_type == T_LEAFLIST) {
                uint32_t j = node.child[0], k = 0;
                do {
                    stack.push_back(j + k);
                    ++k;
                } while (k < node.child[1]);
            } else {
                for (unsigned k = 0; k < 8; ++k)
                    if (node.child[k] != 0u) stack.push_back(node.child[k]);
            }
        }
        acc_out[3 * (i - i_begin)] = a[0];
        acc_out[3 * (i - i_begin) + 1] = a[1];
        acc_out[3 * (i - i_begin) + 2] = a[2];
        if (abs_out) abs_out[i - i_begin] = mag;
    }
}

// src/nbody_gpu_bh.rs:445-527 -- nsteps of the Barnes-Hut binary: build from current positions,
// walk, integrate with clamp, swap.  Returns 0 or a negative error of the stages above.
int orc_step_bh(int law, int64_t n, int nsteps, const float* masses, float G, float eps2, float dt, float theta,
                float world, uint32_t max_depth, int leaflist_mode, float* pos, float* vel, float* acc) {
    std::vector<Node96> nodes((size_t)(4 * n + 64));
    std::vector<float> an((size_t)n * 4, 0.f), po((size_t)n * 4, 0.f), vo((size_t)n * 4, 0.f), ao((size_t)n * 4, 0.f);
    for (int s = 0; s < nsteps; ++s) {
        int64_t cnt = orc_build_octree(0, n, pos, masses, world, max_depth, leaflist_mode, nodes.data(),
                                       (int64_t)nodes.size(), nullptr, nullptr, nullptr);
        if (cnt == -1) {
            nodes.resize(nodes.size() * 4);
            --s;
            continue;
        }
        if (cnt < 0) return (int)cnt;
        int e = orc_bh_accel(law, n, pos, vel, nodes.data(), G, eps2, theta, 0, n, 1 << 20, an.data(), nullptr);
        if (e) return e;
        orc_integrate_wgsl(n, dt, world, pos, vel, acc, an.data(), po.data(), vo.data(), ao.data());
        std::memcpy(pos, po.data(), (size_t)n * 16);
        std::memcpy(vel, vo.data(), (size_t)n * 16);
        std::memcpy(acc, ao.data(), (size_t)n * 16);
    }
    return 0;
}

// Total energy in f64 for the softened (PLUMMER / WGSL) potential: K = sum 0.5 m v^2,
// U = -G sum_{i<j} m_i m_j / sqrt(r^2 + eps2).
void orc_energy(int64_t n, const float* pos16, const float* vel16, const float* masses, double G, double eps2,
                double* kinetic, double* potential) {
    double K = 0, U = 0;
#pragma omp parallel for reduction(+ : K, U) schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        double vx = vel16[4 * i], vy = vel16[4 * i + 1], vz = vel16[4 * i + 2];
        K += 0.5 * masses[i] * (vx * vx + vy * vy + vz * vz);
        for (int64_t j = i + 1; j < n; ++j) {
            double dx = (double)pos16[4 * j] - pos16[4 * i], dy = (double)pos16[4 * j + 1] - pos16[4 * i + 1],
                   dz = (double)pos16[4 * j + 2] - pos16[4 * i + 2];
            U -= G * masses[i] * masses[j] / std::sqrt(dx * dx + dy * dy + dz * dz + eps2);
        }
    }
    *kinetic = K;
    *potential = U;
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

}  // extern "C"
