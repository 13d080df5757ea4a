"""The Rust shim (rust/src/lib.rs) cannot be compiled here (no cargo / rustc in the image), so at least its FFI surface
is checked mechanically: every function in its `unsafe extern "C"` block must be declared in include/rustradio_amd.h
with the same arity and ABI-compatible argument / return types."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# C type -> Rust FFI type (after normalising whitespace, `const`, and parameter names)
C2RUST = {
    "int": "c_int", "float": "f32", "size_t": "usize", "void": "()",
    "const char *": "*const libc::c_char", "const rr_c32 *": "*const Complex", "rr_c32 *": "*mut Complex",
    "const float *": "*const f32", "float *": "*mut f32", "const void *": "*const c_void", "void *": "*mut c_void",
    "size_t *": "*mut usize", "rr_block *": "*mut RrBlock", "const rr_block *": "*const RrBlock",
    "rr_dstream *": "*mut RrDStream", "const rr_dstream *": "*const RrDStream",
    "const void **": "*mut *const c_void", "void **": "*mut *mut c_void", "double *": "*mut f64",
    "unsigned long long *": "*mut u64", "const rr_build_opts *": "*const RrBuildOpts",
    "unsigned long long": "u64", "rr_fanout *": "*mut RrFanout",
}


def _norm_c(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", " *", t)               # `const T *name` -> `const T *name`
    t = re.sub(r"\*\s+\*", "**", t)
    return t


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "rustradio_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^([A-Za-z_][\w \t\*]*?)\b(rr_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = _norm_c(m.group(1)), m.group(2), m.group(3)
        params = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = _norm_c(a)
                a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a     # drop the parameter name
                a = re.sub(r"\*(\w+)$", "*", a).strip()
                params.append(_norm_c(a).replace(" **", " **").replace("* *", "**"))
        protos[name] = (ret, params)
    return protos


def rust_externs():
    src = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    blk = re.search(r'unsafe extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"fn\s+(rr_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if ":" in a]
        out[name] = (ret, params)
    return out


def test_every_rust_extern_matches_the_header():
    c, r = c_prototypes(), rust_externs()
    assert len(r) >= 35, sorted(r)
    for name, (rret, rparams) in r.items():
        assert name in c, f"{name} is bound in rust/src/lib.rs but not declared in include/rustradio_amd.h"
        cret, cparams = c[name]
        assert C2RUST[cret] == rret, (name, cret, rret)
        assert len(cparams) == len(rparams), (name, cparams, rparams)
        for cp, rp in zip(cparams, rparams):
            assert C2RUST[cp] == rp, (name, cp, rp)


def test_every_block_constructor_of_the_header_is_bound():
    c, r = c_prototypes(), rust_externs()
    creates = [n for n in c if n.endswith("_create")]
    assert len(creates) >= 20
    missing = [n for n in creates if n not in r]
    assert not missing, missing
    for n in ("rr_block_work", "rr_block_work_dev", "rr_block_work_streams", "rr_block_eof", "rr_block_destroy",
              "rr_dstream_copy_in", "rr_dstream_copy_out", "rr_host_register", "rr_last_error"):
        assert n in r, n
