// Links librthip.so.  Set RTHIP_LIB_DIR to the directory that holds it
// (raytracer_amd/lib after `python __graft_entry__.py`).
fn main() {
    if let Ok(dir) = std::env::var("RTHIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rthip");
}
