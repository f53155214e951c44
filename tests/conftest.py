import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def data_dir():
    return os.path.join(ROOT, "tests", "data")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it on demand (plain gcc, seconds)."""
    so = os.path.join(ROOT, "oracle", "liborc.so")
    src = os.path.join(ROOT, "oracle", "rb_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    # the product library normally travels prebuilt (see __graft_entry__.build()); build it if a fresh
    # checkout lacks it -- the tests never fall back to anything else
    csrc = os.path.join(ROOT, "rowbowt_amd", "csrc")
    lib = os.path.join(ROOT, "rowbowt_amd", "librbg.so")
    cli = os.path.join(ROOT, "rowbowt_amd", "rb_align")
    sources = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h", ".hpp"))]
    sources += [os.path.join(csrc, "capi", f) for f in os.listdir(os.path.join(csrc, "capi")) if f.endswith(".ipp")]   # the parts of rbg_capi.hip
    newest = max(os.path.getmtime(f) for f in sources)
    cli2 = os.path.join(ROOT, "rowbowt_amd", "rb_markers")
    if not os.path.exists(lib) or not os.path.exists(cli) or not os.path.exists(cli2) or not os.path.exists(os.path.join(ROOT, "rowbowt_amd", "rb_build")) or os.path.getmtime(lib) < newest:
        subprocess.check_call(["make", "-C", csrc, "-j4"])
    yield


# ---- fixtures of the GPU parity files (tests/test_gpu_*.py); nothing here touches a device until a test asks for one --------
@pytest.fixture(scope="session")
def small(data_dir):
    """the reference's toy fixture through rbg_load on device 0, and the oracle on the same files"""
    import orc
    import rowbowt_amd as ra
    rb = ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=0)
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
    yield rb, o
    rb.close()
    o.close()


@pytest.fixture(scope="session")
def simple_reads(data_dir):
    import orc
    return orc.read_fastx(os.path.join(data_dir, "simple_query.fq"))[1]


@pytest.fixture(scope="session")
def error_reads(data_dir):
    import orc
    return orc.read_fastx(os.path.join(data_dir, "error_query.fq"))[1]


@pytest.fixture(scope="session")
def synth():
    """a synthetic pangenome small enough for the oracle and an explicit-text FM index (both position widths, ragged reads)"""
    from synth import SynthIndex
    return SynthIndex(L=4000, H=8, n_sites=60, seed=11)
