"""The product's kernels on the CPU EMULATOR (tests/emu: the .hip sources compiled for the host, every lane of a wavefront a fiber) against the
oracle.  This container has no GPU: the emulator is where the wavefront code is stepped through, asserted on and run under AddressSanitizer
before it goes to the GPU box; results must not depend on the order in which the lanes run between two cross-lane operations
(BVG_EMU_ORDER=rev), or an LDS dependency lacks its wave_sync().  The emulated library is test infrastructure: the product never loads it
(tests/test_abi.py::test_product_never_touches_the_oracle covers tests/ as a whole: nothing under webgraph-big_amd/ or bench.py names it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-j4", "-C", EMU, "libbvgraph_emu.so"])
    return os.path.join(EMU, "libbvgraph_emu.so")


def run_case(*args, **env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    e.pop("BVG_HIP_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(EMU, "run_case.py")] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


@pytest.mark.parametrize("order", ["fwd", "rev"])
def test_sparse_graph_through_the_emulated_kernels(emu_lib, order):
    out = run_case(12000, 3, "web", 3, BVG_EMU_ORDER=order)
    assert "emu case ok" in out and "lean_blocks 0 " in out.splitlines()[0]          # the first scan builds the index on the checking kernels ...
    assert "lean_blocks 0 " not in out.splitlines()[2]                                # ... and the steady state runs the lean kernel


@pytest.mark.parametrize("recs,order,shape,n,seed", [(64, "fwd", "eu", 6000, 5), (64, "rev", "web", 12000, 3), (128, "fwd", "web", 12000, 3), (256, "rev", "eu", 6000, 5), (256, "fwd", "cnr", 40000, 0)])
def test_flat_scan_kernel_on_the_emulator(emu_lib, recs, order, shape, n, seed):
    """experimental/bvg_flat.hip (round 5's flat task kernel: per-record state in an LDS table, run items instead of the position loop) against the oracle: every
    pass structure (64 / 128 / 256 records per super-row), both lane orders, a dense, a sparse and the reference's own graph."""
    out = run_case(n, seed, shape, 2, BVG_FLAT=1, BVG_FLAT_RECS=recs, BVG_EMU_ORDER=order)
    assert "emu case ok" in out and "lean_blocks 0 " not in out.splitlines()[1]


def test_a_failed_allocation_during_the_index_build_is_transient(emu_lib):
    """bvg_api.hip give_up(): an out-of-memory failure of the index build is remembered (no counting pass per scan), announced once, and retried every 8th scan."""
    e = dict(os.environ); e.pop("BVG_HIP_LIB", None); e.pop("BVG_DEBUG", None)
    out = subprocess.run([sys.executable, os.path.join(EMU, "run_oom.py")], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "oom case ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stderr.count("warning: the residual skip index") == 1 and "out of device memory" in out.stderr, out.stderr[-2000:]


@pytest.mark.skipif(not os.environ.get("BVG_EMU_ASAN"), reason="opt-in (BVG_EMU_ASAN=1): the AddressSanitizer build of the emulated library takes ~4 minutes to compile")
@pytest.mark.parametrize("flat", [0, 1])
def test_kernels_under_address_sanitizer(flat):
    """The LDS and global-memory indexing of the row, scan and flat kernels under ASan + UBSan (the dynamic LDS of a launch is a heap block of exactly the bytes the
    launch asked for): round 5 found one out-of-allocation LDS read in rows_kernel this way (harmless on the hardware, fixed)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", EMU, "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    out = run_case(6000, 5, "eu", 2, BVG_EMU_LIB="libbvgraph_emu_asan.so", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", BVG_FLAT=flat, BVG_FLAT_RECS=128)
    assert "emu case ok" in out


def test_malformed_stream_suite_on_the_emulator(emu_lib):
    """tests/test_malformed_streams.py -- hand-assembled records the encoder never writes (equal heads between the three streams, cap at d, over-running copy blocks,
    negative residual counts), every tier and emission mode against the oracle -- is a GPU suite; the emulator runs its 56 GPU cases on the CPU, so the refusal and
    fail-over logic of the kernels is exercised by the driver's CPU run too."""
    e = dict(os.environ, BVG_HIP_LIB=emu_lib, BVG_TEST_KNOBS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_malformed_streams.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                         env=e, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_randomised_parity_of_the_lean_kernels_on_the_emulator(emu_lib):
    """tests/emu/fuzz_flat.py: random shapes, windows, reference-chain depths, interval lengths, zeta k, LDS geometries (small pools: sub-rows and compaction), records per
    super-row and lane orders; scan_kernel and the experimental flat kernel against the oracle (8 cases here; 90 ran on the final round-5 tree: BVG_EMU_FUZZ=<n> for more)."""
    n = os.environ.get("BVG_EMU_FUZZ", "8")
    out = subprocess.run([sys.executable, os.path.join(EMU, "fuzz_flat.py"), n, "5"], capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0 and " 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-1500:]


def test_dense_graph_through_the_emulated_kernels(emu_lib):
    out = run_case(6000, 5, "eu", 3)
    assert "emu case ok" in out and "lean_blocks 0 " not in out.splitlines()[2]


@pytest.mark.parametrize("dbg,order", [(4096, "fwd"), (8192, "rev"), (12288, "fwd")])
def test_round6_list_builds_on_the_emulator(emu_lib, dbg, order):
    """scan_kernel's two round-6 experiments (compiled into the emulated and the experimental library only; both measured slower: csrc/bvg_scan.hip, "MEASURED") stay bit-exact:
    WW (BVG_DBG=4096: stored lists with reference built wave-wide from lane bit vectors) and ZE (8192: position tasks by kept element over the extras' bit vectors)."""
    out = run_case(6000, 5, "eu", 3, BVG_DBG=dbg, BVG_EMU_ORDER=order)
    assert "emu case ok" in out and "lean_blocks 0 " not in out.splitlines()[2]
