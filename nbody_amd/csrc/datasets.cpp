// Synthetic initial conditions (host code).  Stateless counter-based splitmix64 stream so that any
// body range can be generated independently (each shard generates its own bodies) and other
// languages can reproduce the data exactly.
#include <cmath>
#include <cstdint>
#include "../../include/nbody_hip.h"

namespace {

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
// draw k (0..63) of body i: uniform in [0,1) with 24 bits
inline double draw(uint64_t seed, uint64_t i, unsigned k) {
    uint64_t x = mix64(seed + (64ull * i + k + 1ull) * 0x9E3779B97F4A7C15ull);
    return (double)(x >> 40) * (1.0 / 16777216.0);
}

}  // namespace

extern "C" {

// src/nbody_gpu.rs:61-81: mass = U[0.5,8]*U[0.5,8]; position uniform in [world/5, 4*world/5]^3.
int nbody_gen_refcube(uint64_t seed, uint64_t first, uint64_t count, float world, float* host_masses,
                      float* host_pos16) {
    if (!host_masses || !host_pos16) return NBODY_ERR_INVALID;
    const float lo = world / 5.0f;
    const float span = 4.0f * lo - lo;
    for (uint64_t r = 0; r < count; ++r) {
        uint64_t i = first + r;
        float u0 = (float)draw(seed, i, 0), u1 = (float)draw(seed, i, 1), u2 = (float)draw(seed, i, 2);
        float u3 = (float)draw(seed, i, 3), u4 = (float)draw(seed, i, 4);
        host_pos16[4 * r + 0] = lo + span * u0;
        host_pos16[4 * r + 1] = lo + span * u1;
        host_pos16[4 * r + 2] = lo + span * u2;
        host_pos16[4 * r + 3] = 0.0f;
        host_masses[r] = (0.5f + 7.5f * u3) * (0.5f + 7.5f * u4);
    }
    return NBODY_OK;
}

int nbody_gen_plummer(uint64_t seed, uint64_t first, uint64_t count, uint64_t n_total, float world, float a,
                      float rmax, float total_mass, float G, int equilibrium, float* host_masses, float* host_pos16,
                      float* host_vel16) {
    if (!host_masses || !host_pos16 || n_total == 0) return NBODY_ERR_INVALID;
    const double c = 0.5 * (double)world;
    const double two_pi = 6.283185307179586476925286766559;
    const float m = total_mass / (float)n_total;
    for (uint64_t r = 0; r < count; ++r) {
        uint64_t i = first + r;
        // radius by inversion of the cumulative mass, rejected beyond rmax (draws 0..19)
        double rad = (double)a;  // fallback if every attempt is rejected
        for (unsigned k = 0; k < 20; ++k) {
            double u = draw(seed, i, k);
            if (u <= 0.0) continue;
            double t = std::pow(u, -2.0 / 3.0) - 1.0;
            if (t <= 0.0) continue;
            double cand = (double)a / std::sqrt(t);
            if (cand <= (double)rmax) {
                rad = cand;
                break;
            }
        }
        double z = 1.0 - 2.0 * draw(seed, i, 20);
        double phi = two_pi * draw(seed, i, 21);
        double s = std::sqrt(std::fmax(0.0, 1.0 - z * z));
        host_pos16[4 * r + 0] = (float)(c + rad * s * std::cos(phi));
        host_pos16[4 * r + 1] = (float)(c + rad * s * std::sin(phi));
        host_pos16[4 * r + 2] = (float)(c + rad * z);
        host_pos16[4 * r + 3] = 0.0f;
        host_masses[r] = m;
        if (host_vel16) {
            double vx = 0, vy = 0, vz = 0;
            if (equilibrium) {
                // Aarseth, Henon & Wielen (1974): q = v/v_esc with density q^2 (1-q^2)^(7/2),
                // von Neumann rejection (draws 22..59 in pairs), isotropic direction (60, 61).
                double q = 0.5;
                for (unsigned k = 22; k + 1 < 60; k += 2) {
                    double x = draw(seed, i, k), y = 0.1 * draw(seed, i, k + 1);
                    if (y < x * x * std::pow(1.0 - x * x, 3.5)) {
                        q = x;
                        break;
                    }
                }
                double vesc = std::sqrt(2.0 * (double)G * (double)total_mass / (double)a) *
                              std::pow(1.0 + (rad * rad) / ((double)a * (double)a), -0.25);
                double v = q * vesc;
                double vzc = 1.0 - 2.0 * draw(seed, i, 60);
                double vphi = two_pi * draw(seed, i, 61);
                double vs = std::sqrt(std::fmax(0.0, 1.0 - vzc * vzc));
                vx = v * vs * std::cos(vphi);
                vy = v * vs * std::sin(vphi);
                vz = v * vzc;
            }
            host_vel16[4 * r + 0] = (float)vx;
            host_vel16[4 * r + 1] = (float)vy;
            host_vel16[4 * r + 2] = (float)vz;
            host_vel16[4 * r + 3] = 0.0f;
        }
    }
    return NBODY_OK;
}

}  // extern "C"
