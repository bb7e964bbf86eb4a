// UNVERIFIED (no Rust toolchain in the build image).
fn main() {
    let dir = std::env::var("MVF_GPU_LIB_DIR").unwrap_or_else(|_| "../../metrovector_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mvf_gpu");
    println!("cargo:rerun-if-env-changed=MVF_GPU_LIB_DIR");
}
