// Links libstringwars_amd.so, built in-tree by `make -C stringwars_amd/csrc` (one directory above this crate).
// STRINGWARS_AMD_LIB_DIR overrides the search path; the run-time loader finds the library through the rpath.
fn main() {
    let manifest = std::env::var("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR");
    let dir = std::env::var("STRINGWARS_AMD_LIB_DIR").unwrap_or_else(|_| format!("{}/..", manifest));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=stringwars_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=STRINGWARS_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/stringwars_amd.h");
}
