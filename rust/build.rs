// Link against librustradio_amd.so built by `make -C rustradio_amd/csrc` (hipcc, gfx950).
fn main() {
    let dir = std::env::var("RUSTRADIO_AMD_LIB_DIR").unwrap_or_else(|_| "../rustradio_amd/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rustradio_amd");
    println!("cargo:rerun-if-env-changed=RUSTRADIO_AMD_LIB_DIR");
}
