//! The step blocks of the reference's three binaries over libnbody_hip.so, without window, camera
//! or renderer: `--mode gpu` (src/nbody_gpu.rs:418-451), `--mode bh` (src/nbody_gpu_bh.rs:445-527),
//! `--mode cpu` (src/nbody_cpu.rs:319-360).  Each frame is one step; every `--report` frames the
//! positions come back to the host, as the Barnes-Hut binary reads them every frame.
use nbody_hip_sys as nb;
use std::time::Instant;

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let mut mode = String::from("gpu");
    let (mut n, mut frames, mut report, mut seed) = (250u64, 100i32, 50i32, 42u64); // N_BODIES, src/nbody_gpu.rs:14
    let mut args = std::env::args().skip(1);
    while let Some(a) = args.next() {
        let mut value = || args.next().unwrap_or_default();
        match a.as_str() {
            "--mode" => mode = value(),
            "--n" => n = value().parse()?,
            "--frames" => frames = value().parse()?,
            "--report" => report = value().parse()?,
            "--seed" => seed = value().parse()?,
            _ => {
                eprintln!("usage: nbody-headless [--mode cpu|gpu|bh] [--n N] [--frames F] [--report R] [--seed S]");
                std::process::exit(2);
            }
        }
    }
    let mut cfg = nb::default_config(n);
    if mode == "cpu" {
        cfg.law = nb::NBODY_LAW_CPU_R6;
        cfg.integrator = nb::NBODY_INT_CPU_EULER_XBUF;
    }
    let mut masses = vec![0f32; n as usize];
    let mut pos = vec![0f32; 4 * n as usize];
    // initial conditions of src/nbody_gpu.rs:61-81
    let rc = unsafe { nb::nbody_gen_refcube(seed, 0, n, cfg.world, masses.as_mut_ptr(), pos.as_mut_ptr()) };
    assert_eq!(rc, nb::NBODY_OK);
    let mut ctx = nb::Context::new(&cfg)?;
    ctx.upload(&masses, &pos, None)?;
    let t0 = Instant::now();
    for f in 1..=frames {
        if mode == "bh" {
            ctx.step_bh(1)?;
        } else {
            ctx.step_allpairs(1)?;
        }
        if report > 0 && f % report == 0 {
            ctx.download_positions(&mut pos)?;
            println!("frame {}: body0 = ({:e}, {:e}, {:e})", f, pos[0], pos[1], pos[2]);
        }
    }
    ctx.sync()?;
    let s = t0.elapsed().as_secs_f64();
    let c = ctx.counters()?;
    println!(
        "{}: n={} frames={}  {:.3} ms/frame  {:.1} frames/s  octree nodes (last) {}",
        mode, n, frames, 1e3 * s / frames as f64, frames as f64 / s, c.octree_nodes
    );
    Ok(())
}
