// build.rs -- link the prebuilt libnenbody_hip.so (built by `make -C nenbody_amd/csrc`, hipcc, gfx950).
// NENBODY_HIP_LIB_DIR must point at the directory holding it (nenbody_amd/lib in this repository).
fn main() {
    let dir = std::env::var("NENBODY_HIP_LIB_DIR").expect("set NENBODY_HIP_LIB_DIR to the directory of libnenbody_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=nenbody_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=NENBODY_HIP_LIB_DIR");
}
