// Links libfecgpu.so.  FECGPU_LIB_DIR = the directory that holds it (forge_ec_amd/ of the MI355X
// repository after `python -m forge_ec_amd.build`).
fn main() {
    if let Ok(dir) = std::env::var("FECGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=fecgpu");
    println!("cargo:rerun-if-env-changed=FECGPU_LIB_DIR");
}
