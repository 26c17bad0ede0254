"""CPU only (never on the GPU box): the host-side code of the product under AddressSanitizer + UndefinedBehaviorSanitizer
and, for the worker pool, ThreadSanitizer (SURVEY.md 5 "race detection / sanitizers"; the reference's only threads are
parFor's, lib/gpc/filter.hpp:131-141; its PNG reader is lib/gpc/buffer.hpp:197-318).

What runs: the forest parser on truncated / non-numeric / over-long files, gpc_hip_expand_packed with ragged capacities
and alignments, the expansion pool as gpc_hip_match_batch drives it (1-8 workers, copies and expansions mixed), the PNG
decoder on every truncation of a valid file, forged headers and non-existent filter types -- and the oracle's C code on
the golden cases.  None of it needs a device; a sanitizer report fails the test."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "opengpc_amd", "csrc", "gpc_hip.hip")
DRV = os.path.join(ROOT, "tests", "cpp", "sanitize_host.cpp")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None or not os.path.exists(CLANG), reason="needs the ROCm toolchain")


def build(tmp, san, tag, png=True):
    lib = os.path.join(tmp, "libgpc_hip_%s.so" % tag)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fno-omit-frame-pointer",
                           "-fsanitize=" + san, "-fno-gpu-sanitize", "-o", lib, SRC], stderr=subprocess.DEVNULL)
    exe = os.path.join(tmp, "sanitize_host_" + tag)
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + san,
                           "-I" + os.path.join(ROOT, "include")] + ([] if png else ["-DNO_PNG"]) +
                          ["-o", exe, DRV, lib, "-lz", "-lpthread", "-Wl,-rpath," + tmp, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, *args, env=None):
    e = dict(os.environ)
    # (the HIP runtime the library links keeps allocations of its own alive at exit: leaks are not what is looked for)
    e.update(ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
             TSAN_OPTIONS="halt_on_error=1:report_signal_unsafe=0")
    e.update(env or {})
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


@pytest.fixture(scope="module")
def asan_exe(tmp_path_factory):
    return build(str(tmp_path_factory.mktemp("asan")), "address,undefined", "asan")


def test_forest_parser_under_asan_ubsan(asan_exe, forest_paths):
    for name in ("zero", "tau"):
        out = run(asan_exe, "forest", forest_paths[name])
        assert "OK forest whole status 0 tests 30" in out and "OK forest 16x20 status 0 tests 32 discarded 288" in out
        assert "OK forest nasty texts" in out


def test_expansion_of_packed_results_under_asan_ubsan(asan_exe):
    assert "OK expand" in run(asan_exe, "expand")
    assert "OK pool" in run(asan_exe, "pool")


def test_resident_image_fingerprint_under_asan_ubsan(asan_exe):
    """the fingerprint a resident image's host copies are held against (gpc_hip.hip: fingerprint, ranges_overlap): exact-size
    heap arrays from 1 byte up, sampled and full"""
    out = run(asan_exe, "fingerprint")
    assert "OK fingerprint 126 array sets" in out


def test_png_decoder_under_asan_ubsan(asan_exe, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    Image.fromarray(rng.integers(0, 255, (30, 40), dtype=np.uint8)).save(tmp_path / "good.png")
    out = run(asan_exe, "png", str(tmp_path))
    # (a file cut inside its 12-byte IEND chunk still holds every pixel: those truncations decode, all others are refused)
    acc = int(out.split("OK png truncations refused")[1].split("accepted")[1].split()[0])
    assert acc <= 12
    assert "OK png forged headers" in out and "OK png filters" in out


def test_expansion_pool_under_tsan(tmp_path_factory):
    exe = build(str(tmp_path_factory.mktemp("tsan")), "thread", "tsan", png=False)
    assert "OK pool" in run(exe, "pool")


def test_oracle_under_asan_ubsan(tmp_path, golden, forest_paths):
    """The checker's own C code (oracle/gpc_oracle.c) on the Appendix-C cases, in a child python with the ASan runtime preloaded."""
    lib = tmp_path / "libgpc_oracle_asan.so"
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-ffp-contract=off",
                           "-fPIC", "-shared", "-o", str(lib), os.path.join(ROOT, "oracle", "gpc_oracle.c"),
                           os.path.join(ROOT, "oracle", "gpc_oracle_train.c")])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import oracle.pyoracle as po\n"
            "po.ORACLE_SO = %r\n"
            "o = po.Oracle()\n"
            "for (W, H, D) in ((96, 64, 5), (160, 100, 7), (1024, 436, 24)):\n"
            "    L, R = o.synth_pair(W, H, 0, D)\n"
            "    for f in (%r, %r):\n"
            "        rc, fm = o.read_forest(f, W, H)\n"
            "        for ep in (True, False):\n"
            "            s = po.sparsematch_settings(); s.epipolar_mode = int(ep)\n"
            "            supp, nl, nr = o.match_pair(L, R, fm, s)\n"
            "            print('CASE', W, H, ep, len(supp), nl, nr)\n"
            % (ROOT, str(lib), forest_paths["zero"], forest_paths["tau"]))
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
    cases = [l.split() for l in r.stdout.splitlines() if l.startswith("CASE")]
    assert len(cases) == 12
    # the 1024x436 Zero epipolar case of SURVEY.md Appendix C: 269 547 supports, 285 209 / 285 139 candidates
    assert ["CASE", "1024", "436", "True", "269547", "285209", "285139"] in cases
