// Links libnbody_hip.so.  NBODY_HIP_LIB_DIR names the directory that holds it (the repository's
// nbody_amd/ after `make -C nbody_amd/csrc`).
fn main() {
    println!("cargo:rerun-if-env-changed=NBODY_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("NBODY_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=nbody_hip");
}
