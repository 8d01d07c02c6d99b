"""What the compiler emitted for gfx950 (hipcc cross-compiles without a GPU):

* bit-exactness rests on strict IEEE evaluation in the reference's order, so no kernel may
  contain a fused multiply-add outside the expansion of an fp64 DIVISION (which is an IEEE
  division whatever instructions implement it; only the extension's coarsest-grid solver
  divides on the device);
* GPU code must not store or do atomics through the scalar data cache, nor write it back
  or discard it.
"""
import os
import re
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "multigrid_amd", "csrc")
SOURCES = ["mgx_kernels.hip", "mgx_stream.hip", "mgx_tail.hip", "mgx_coarse.hip", "mgx_noalloc.hip", "mgx_mid.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]


def _isa(tmp_path):
    procs = []
    for src in SOURCES:
        out = tmp_path / (src + ".s")
        procs.append((src, out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", str(out)],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    res = {}
    for src, out, p in procs:
        log, _ = p.communicate(timeout=900)
        assert p.returncode == 0, log
        res[src] = out.read_text().splitlines()
    return res


def test_flags_match_the_library_build():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-ffp-contract=off" in mk and "-fno-fast-math" in mk and "gfx950" in mk


def test_no_fma_outside_divisions_and_no_scalar_cache_writes(tmp_path):
    isa = _isa(tmp_path)
    fma = re.compile(r"\bv_(fma|fmac|mad|pk_fma)_f(64|32)\b")
    scalar_write = re.compile(r"\b(s_store|s_buffer_store|s_atomic|s_buffer_atomic|s_dcache_wb|s_dcache_discard|s_scratch_store)")
    for src, lines in isa.items():
        in_div = False
        stray = []
        nmul = 0
        for ln in lines:
            t = ln.strip()
            if t.startswith("v_div_scale_f64"):
                in_div = True
            elif t.startswith("v_div_fixup_f64"):
                in_div = False
            elif fma.search(t) and not in_div:
                stray.append(t)
            if t.startswith("v_mul_f64") or t.startswith("v_add_f64"):
                nmul += 1
            assert not scalar_write.search(t), "%s: %s" % (src, t)
        assert not stray, "%s: fused multiply-add outside a division: %s" % (src, stray[:3])
        assert nmul >= (5 if src == "mgx_noalloc.hip" else 20), "%s: no fp64 arithmetic found - wrong file?" % src
    # the bit-exact kernels do not divide at all
    for src in ("mgx_kernels.hip", "mgx_stream.hip"):
        assert not any("v_div_scale_f64" in ln for ln in isa[src]), src
