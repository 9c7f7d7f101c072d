//! Raw bindings of `include/nbody_hip.h` and a thin safe wrapper.
//!
//! Buffer layouts are the reference's: `f32[N]` masses; positions, velocities and accelerations as
//! `f32[N][4]` (WGSL `vec3` at a 16-byte stride); 96-byte octree nodes.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const NBODY_OK: c_int = 0;
pub const NBODY_ERR_INVALID: c_int = -1;
pub const NBODY_ERR_HIP: c_int = -2;
pub const NBODY_ERR_NO_DEVICE: c_int = -3;
pub const NBODY_ERR_STATE: c_int = -4;
pub const NBODY_ERR_CAPACITY: c_int = -5;
pub const NBODY_ERR_NOMEM: c_int = -6;

pub const NBODY_LAW_CPU_R6: i32 = 0;
pub const NBODY_LAW_WGSL_BIAS: i32 = 1;
pub const NBODY_LAW_PLUMMER: i32 = 2;

pub const NBODY_INT_CPU_EULER_XBUF: i32 = 0;
pub const NBODY_INT_WGSL_VERLET: i32 = 1;
pub const NBODY_INT_LEAPFROG: i32 = 2;

pub const NBODY_NODE_DUMMY: u32 = 0;
pub const NBODY_NODE_LEAFBODY: u32 = 1;
pub const NBODY_NODE_LEAFLIST: u32 = 2;
pub const NBODY_NODE_INTERIOR: u32 = 3;

pub const NBODY_BUF_POSM: c_int = 0;
pub const NBODY_BUF_VEL: c_int = 1;
pub const NBODY_BUF_ACC: c_int = 2;
pub const NBODY_BUF_OCTREE: c_int = 3;
pub const NBODY_BUF_ACC_PARTIAL: c_int = 4;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct nbody_config {
    pub n_bodies: u64,
    pub law: i32,
    pub integrator: i32,
    pub g: f32,
    pub dt: f32,
    pub eps2: f32,
    pub theta: f32,
    pub world: f32,
    pub max_depth: u32,
    pub leaflist_mode: i32,
    pub shard_begin: u64,
    pub shard_count: u64,
    pub device: i32,
    pub quadrupole: i32,
    pub stream: *mut c_void,
}

/// 96 bytes, the storage layout of `src/nbodybh.wgsl:1-9`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct nbody_octree_node {
    pub center_of_mass: [f32; 3],
    pub _pad0: f32,
    pub pos_min: [f32; 3],
    pub _pad1: f32,
    pub pos_max: [f32; 3],
    pub range: f32,
    pub total_mass: f32,
    pub child_indices: [u32; 8],
    pub node_type: u32,
    pub depth: u32,
    pub _pad2: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct nbody_counters_t {
    pub steps_allpairs: u64,
    pub steps_bh: u64,
    pub interactions_allpairs: u64,
    pub bh_node_visits: u64,
    pub bh_leaf_interactions: u64,
    pub bh_aggregate_interactions: u64,
    pub bh_wave_node_visits: u64,
    pub octree_nodes: u64,
    pub bh_max_stack: u64,
    pub allpairs_path: u64,
    pub reserved: [u64; 6],
}

/// Sizes and times of the last domain-decomposed build (`nbody_bh_set_domains`).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct nbody_domain_stats_t {
    pub domains: u32,
    pub top_levels: u32,
    pub top_entries: u64,
    pub top_nodes: u64,
    pub merged_nodes: u64,
    pub top_nodes_visited: u64,
    pub partition_ms: f64,
    pub merge_ms: f64,
    pub bodies: [u64; 32],
    pub nodes: [u64; 32],
    pub imported_nodes: [u64; 32],
    pub build_ms: [f64; 32],
    pub view_domain: i32,
    pub distributed: i32,
    pub exported_nodes: [u64; 32],
    pub export_ms: [f64; 32],
}

#[repr(C)]
pub struct nbody_ctx {
    _private: [u8; 0],
}

const _: () = assert!(std::mem::size_of::<nbody_config>() == 80);
const _: () = assert!(std::mem::size_of::<nbody_octree_node>() == 96);
const _: () = assert!(std::mem::size_of::<nbody_counters_t>() == 128);

extern "C" {
    pub fn nbody_version() -> *const c_char;
    pub fn nbody_last_error(ctx: *const nbody_ctx) -> *const c_char;
    pub fn nbody_config_default(cfg: *mut nbody_config);
    pub fn nbody_create(out: *mut *mut nbody_ctx, cfg: *const nbody_config) -> c_int;
    pub fn nbody_destroy(ctx: *mut nbody_ctx);
    pub fn nbody_upload(ctx: *mut nbody_ctx, host_masses: *const f32, host_pos16: *const f32, host_vel16: *const f32, host_acc16: *const f32) -> c_int;
    pub fn nbody_upload_alt(ctx: *mut nbody_ctx, host_pos16: *const f32, host_vel16: *const f32, host_acc16: *const f32) -> c_int;
    pub fn nbody_step_allpairs(ctx: *mut nbody_ctx, nsteps: c_int) -> c_int;
    pub fn nbody_step_bh(ctx: *mut nbody_ctx, nsteps: c_int) -> c_int;
    pub fn nbody_allpairs_partial(ctx: *mut nbody_ctx, part: u32, nparts: u32) -> c_int;
    pub fn nbody_integrate_partial(ctx: *mut nbody_ctx) -> c_int;
    pub fn nbody_bh_forces_ranked(ctx: *mut nbody_ctx, rank_begin: u64, rank_count: u64) -> c_int;
    pub fn nbody_integrate_ranked(ctx: *mut nbody_ctx) -> c_int;
    /// No counterpart in the reference: RCCL communicator inside the library (one process per GPU).
    pub fn nbody_comm_unique_id(id_out128: *mut c_void) -> c_int;
    pub fn nbody_comm_init(ctx: *mut nbody_ctx, unique_id128: *const c_void, rank: c_int, world: c_int) -> c_int;
    pub fn nbody_bh_set_domains(ctx: *mut nbody_ctx, domains: u32, top_levels: u32) -> c_int;
    pub fn nbody_bh_domain_stats(ctx: *mut nbody_ctx, out: *mut nbody_domain_stats_t) -> c_int;
    pub fn nbody_bh_set_domain_view(ctx: *mut nbody_ctx, domain: i32) -> c_int;
    pub fn nbody_bh_set_target_order(ctx: *mut nbody_ctx, host_order: *const u32, count: u64) -> c_int;
    pub fn nbody_comm_sorted_run(n_bodies: u64, world: c_int, rank: c_int, begin: *mut u64, count: *mut u64) -> c_int;
    pub fn nbody_comm_exchange_plan(n_bodies: u64, world: c_int, begin: *const u64, count: *const u64) -> c_int;
    pub fn nbody_accel_allpairs(ctx: *mut nbody_ctx, host_acc16_out: *mut f32) -> c_int;
    /// No counterpart in the reference: for callers that cannot allocate device memory.
    pub fn nbody_accel_allpairs_noalloc(dev_posm: *const f32, dev_vel16: *const f32, n_bodies: u64, law: c_int, g: f32, eps2: f32, dev_acc16_out: *mut f32, stream: *mut c_void) -> c_int;
    pub fn nbody_accel_bh(ctx: *mut nbody_ctx, host_acc16_out: *mut f32) -> c_int;
    pub fn nbody_build_octree(ctx: *mut nbody_ctx) -> c_int;
    pub fn nbody_download_octree(ctx: *mut nbody_ctx, host_nodes: *mut nbody_octree_node, cap_nodes: u64, n_nodes: *mut u64) -> c_int;
    pub fn nbody_upload_octree(ctx: *mut nbody_ctx, host_nodes: *const nbody_octree_node, n_nodes: u64) -> c_int;
    pub fn nbody_morton_keys(ctx: *mut nbody_ctx, host_keys: *mut u64, host_leaf_depth: *mut u8, host_order: *mut u32) -> c_int;
    pub fn nbody_download(ctx: *mut nbody_ctx, host_pos16: *mut f32, host_vel16: *mut f32, host_acc16: *mut f32) -> c_int;
    pub fn nbody_download_alt(ctx: *mut nbody_ctx, host_pos16: *mut f32, host_vel16: *mut f32, host_acc16: *mut f32) -> c_int;
    pub fn nbody_sync(ctx: *mut nbody_ctx) -> c_int;
    pub fn nbody_energy(ctx: *mut nbody_ctx, kinetic: *mut f64, potential: *mut f64) -> c_int;
    pub fn nbody_counters(ctx: *mut nbody_ctx, out: *mut nbody_counters_t) -> c_int;
    pub fn nbody_counters_reset(ctx: *mut nbody_ctx) -> c_int;
    pub fn nbody_set_counting(ctx: *mut nbody_ctx, on: c_int) -> c_int;
    pub fn nbody_device_buffer(ctx: *mut nbody_ctx, which: c_int, dev_ptr: *mut *mut c_void, bytes: *mut usize) -> c_int;
    pub fn nbody_read_buffer(ctx: *mut nbody_ctx, which: c_int, host_dst: *mut c_void, bytes: usize) -> c_int;
    pub fn nbody_stream(ctx: *mut nbody_ctx) -> *mut c_void;
    pub fn nbody_kernel_time(ctx: *mut nbody_ctx, which: c_int, avg_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn nbody_set_kernel_timing(ctx: *mut nbody_ctx, on: c_int) -> c_int;
    pub fn nbody_kernel_clock(ctx: *mut nbody_ctx, mhz: *mut f64) -> c_int;
    pub fn nbody_gen_refcube(seed: u64, first: u64, count: u64, world: f32, host_masses: *mut f32, host_pos16: *mut f32) -> c_int;
    pub fn nbody_gen_plummer(seed: u64, first: u64, count: u64, n_total: u64, world: f32, a: f32, rmax: f32, total_mass: f32, g: f32, equilibrium: c_int, host_masses: *mut f32, host_pos16: *mut f32, host_vel16: *mut f32) -> c_int;
}

/// An error code of the library with its message.
#[derive(Debug)]
pub struct Error {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "nbody_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

/// The reference's constants (`nbody_config_default`) for `n_bodies` bodies.
pub fn default_config(n_bodies: u64) -> nbody_config {
    let mut cfg = std::mem::MaybeUninit::<nbody_config>::uninit();
    // SAFETY: nbody_config_default writes every field.
    let mut cfg = unsafe {
        nbody_config_default(cfg.as_mut_ptr());
        cfg.assume_init()
    };
    cfg.n_bodies = n_bodies;
    cfg
}

/// One simulation context on one GPU.  Single-threaded, like the reference's event loop.
pub struct Context {
    raw: *mut nbody_ctx,
    n: usize,
}

impl Context {
    pub fn new(cfg: &nbody_config) -> Result<Self, Error> {
        let mut raw: *mut nbody_ctx = std::ptr::null_mut();
        // SAFETY: both pointers are valid for the call.
        let rc = unsafe { nbody_create(&mut raw, cfg) };
        if rc != NBODY_OK {
            return Err(Error { code: rc, message: last_error(std::ptr::null()) });
        }
        Ok(Context { raw, n: cfg.n_bodies as usize })
    }

    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == NBODY_OK {
            Ok(())
        } else {
            Err(Error { code: rc, message: last_error(self.raw) })
        }
    }

    pub fn n_bodies(&self) -> usize {
        self.n
    }

    /// `masses`: N values; `pos16` (and `vel16` if given): 4 N values, 16-byte stride.
    pub fn upload(&mut self, masses: &[f32], pos16: &[f32], vel16: Option<&[f32]>) -> Result<(), Error> {
        assert!(masses.len() == self.n && pos16.len() == 4 * self.n);
        let vel = match vel16 {
            Some(v) => {
                assert!(v.len() == 4 * self.n);
                v.as_ptr()
            }
            None => std::ptr::null(),
        };
        // SAFETY: lengths checked above; the library copies before returning.
        let rc = unsafe { nbody_upload(self.raw, masses.as_ptr(), pos16.as_ptr(), vel, std::ptr::null()) };
        self.check(rc)
    }

    /// One `MainEventsCleared` tick of `src/nbody_gpu.rs:418-451` (or `src/nbody_cpu.rs:319-360`), `nsteps` times.
    pub fn step_allpairs(&mut self, nsteps: i32) -> Result<(), Error> {
        let rc = unsafe { nbody_step_allpairs(self.raw, nsteps) };
        self.check(rc)
    }

    /// One tick of `src/nbody_gpu_bh.rs:445-527`, `nsteps` times.
    pub fn step_bh(&mut self, nsteps: i32) -> Result<(), Error> {
        let rc = unsafe { nbody_step_bh(self.raw, nsteps) };
        self.check(rc)
    }

    /// Positions as the renderer binds them: `f32[N][4]`, pad lane zero.
    pub fn download_positions(&mut self, pos16: &mut [f32]) -> Result<(), Error> {
        assert!(pos16.len() == 4 * self.n);
        let rc = unsafe { nbody_download(self.raw, pos16.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) };
        self.check(rc)
    }

    pub fn download_octree(&mut self) -> Result<Vec<nbody_octree_node>, Error> {
        let mut count: u64 = 0;
        let rc = unsafe { nbody_download_octree(self.raw, std::ptr::null_mut(), 0, &mut count) };
        self.check(rc)?;
        let mut nodes = vec![nbody_octree_node::default(); count as usize];
        let rc = unsafe { nbody_download_octree(self.raw, nodes.as_mut_ptr(), count, &mut count) };
        self.check(rc)?;
        Ok(nodes)
    }

    pub fn counters(&mut self) -> Result<nbody_counters_t, Error> {
        let mut c = nbody_counters_t::default();
        let rc = unsafe { nbody_counters(self.raw, &mut c) };
        self.check(rc)?;
        Ok(c)
    }

    pub fn sync(&mut self) -> Result<(), Error> {
        let rc = unsafe { nbody_sync(self.raw) };
        self.check(rc)
    }

    /// For the entry points without a wrapper.
    pub fn as_raw(&mut self) -> *mut nbody_ctx {
        self.raw
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        // SAFETY: raw came from nbody_create and is destroyed once.
        unsafe { nbody_destroy(self.raw) }
    }
}

fn last_error(ctx: *const nbody_ctx) -> String {
    // SAFETY: the library returns a NUL-terminated string that outlives this call.
    unsafe { CStr::from_ptr(nbody_last_error(ctx)) }.to_string_lossy().into_owned()
}
