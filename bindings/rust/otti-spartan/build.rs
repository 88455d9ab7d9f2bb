// Link against otti_amd/libottispartan.so.  OTTI_SPARTAN_LIB_DIR overrides the default location (this repository's otti_amd/).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("OTTI_SPARTAN_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../otti_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ottispartan");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=OTTI_SPARTAN_LIB_DIR");
}
