//! Link against libmadsim_hip.so.  The library is built by `make -C madsim_amd/csrc` (hipcc, gfx950); this script only tells
//! rustc where it is: MADSIM_HIP_LIB_DIR, or <repo>/madsim_amd relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=MADSIM_HIP_LIB_DIR");
    let dir = env::var_os("MADSIM_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        let manifest = PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR"));
        manifest.join("..").join("..").join("..").join("madsim_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    if !dir.join("libmadsim_hip.so").exists() {
        println!(
            "cargo:warning=libmadsim_hip.so not found in {} (set MADSIM_HIP_LIB_DIR or run `make -C madsim_amd/csrc`); there is no CPU fallback",
            dir.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=madsim_hip");
    // dependants can embed the directory as an rpath: DEP_MADSIM_HIP_LIBDIR
    println!("cargo:libdir={}", dir.display());
}
