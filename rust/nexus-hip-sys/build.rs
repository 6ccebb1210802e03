// libnexus_hip.so is built by `make -C nexus-zkvm_amd/csrc` (hipcc, gfx950); point NEXUS_HIP_LIB_DIR at the directory holding it.
fn main() {
    let dir = std::env::var("NEXUS_HIP_LIB_DIR").expect("set NEXUS_HIP_LIB_DIR to the directory of libnexus_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=nexus_hip");
    println!("cargo:rerun-if-env-changed=NEXUS_HIP_LIB_DIR");
}
