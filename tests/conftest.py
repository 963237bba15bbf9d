import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# Round 3 ended with ONE unexplained core dump of the test process in about thirty whole-suite runs, and nothing but the
# interpreter's banner to go by. Every test process now (i) asks the library for a native back-trace on SIGSEGV / SIGBUS /
# SIGABRT / SIGFPE (PG_NATIVE_BACKTRACE, installed when libpagraph_hip.so is loaded — before faulthandler, which chains to it)
# and (ii) keeps Python's faulthandler output of every thread in a file that survives the run (gpurun_out/ travels back from
# the GPU box), besides pytest's own copy on stderr.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")          # as pagraph_amd/__init__.py sets it (before the first HIP call)
os.environ.setdefault("PG_NATIVE_BACKTRACE", "1")
_out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(_out) and os.access(_out, os.W_OK):
    # a copy of that back-trace outside pytest's capture of stderr (which dies with the process). Written only on a fatal signal.
    os.environ.setdefault("PG_NATIVE_BACKTRACE_FILE", os.path.join(_out, f"native_backtrace_{os.getpid()}.txt"))
os.environ.setdefault("PYTHONFAULTHANDLER", "1")        # child processes (mp.spawn workers, subprocess CLIs) too
_FAULT_LOG = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "bounds_selftest: provokes the debug library's bounds record on purpose")


def pytest_sessionstart(session):
    global _FAULT_LOG
    import faulthandler
    out_dir = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(out_dir) or not os.access(out_dir, os.W_OK):
        import tempfile
        out_dir = tempfile.gettempdir()
    try:
        _FAULT_LOG = open(os.path.join(out_dir, f"faulthandler_{os.getpid()}.log"), "w")
        faulthandler.enable(file=_FAULT_LOG, all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=True)


def pytest_sessionfinish(session, exitstatus):
    global _FAULT_LOG
    bt = os.environ.get("PG_NATIVE_BACKTRACE_FILE")
    try:
        if bt and os.path.exists(bt) and os.path.getsize(bt) == 0:
            os.unlink(bt)                     # opened when the library loaded, written only on a fatal signal
    except OSError:
        pass
    if _FAULT_LOG is not None:
        import faulthandler
        faulthandler.disable()
        name = _FAULT_LOG.name
        _FAULT_LOG.close()
        _FAULT_LOG = None
        try:
            if os.path.getsize(name) == 0:
                os.unlink(name)               # nothing happened: leave no litter
        except OSError:
            pass


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A failing GPU test's exception is written to stderr (and the fault log) AT ONCE, with the test's id: the one core dump
    this suite has produced so far (round 4: 1 of ~35 whole-suite runs) happened while pytest was still rendering the failure
    — a sticky `hipErrorIllegalAddress` made torch's ~CUDAGraph throw during a garbage collection inside the report
    generation, std::terminate — so the failure's own text never reached the log."""
    outcome = yield
    if outcome.excinfo is not None:
        import sys
        import traceback
        etype, evalue, tb = outcome.excinfo
        if etype.__name__ not in ("Skipped", "XFailed"):
            msg = f"\n[conftest] {item.nodeid} raised {etype.__name__}: {str(evalue)[:1500]}\n" + \
                  "".join(traceback.format_tb(tb)[-6:])
            sys.stderr.write(msg)
            sys.stderr.flush()
            if _FAULT_LOG is not None:
                try:
                    _FAULT_LOG.write(msg)
                    _FAULT_LOG.flush()
                except (OSError, ValueError):
                    pass


@pytest.fixture(autouse=True)
def _bounds_after_each_test(request):
    """PG_BOUNDS=1 (the debug library that bounds-checks ids, include/pagraph_hip.h pg_bounds_*): after every GPU test the
    library is asked whether a kernel met an out-of-range index; the test then FAILS with the kernel, the call site, the
    value and the bound — instead of a hipErrorIllegalAddress somewhere behind it (tools/hunt_lifetimes.sh)."""
    yield
    if os.environ.get("PG_HUNT_SYNC") and request.node.get_closest_marker("gpu") is not None:
        # tools/hunt_lifetimes.sh: a device-wide wait behind every test, so that an asynchronous fault is reported in the
        # test whose work (or whose teardown) caused it, not at the first synchronising call of a later one
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    if os.environ.get("PG_BOUNDS") in (None, "", "0") or request.node.get_closest_marker("gpu") is None:
        return
    from pagraph_amd import _lib
    if _lib._lib is None or not _lib.BOUNDS or request.node.get_closest_marker("bounds_selftest") is not None:
        return
    rec = _lib.bounds_report(reset=True)
    if rec is not None:
        msg = f"\n[bounds] {request.node.nodeid}: {rec}\n"
        sys.stderr.write(msg)
        sys.stderr.flush()
        pytest.fail(f"PG_BOUNDS: a kernel followed an out-of-range index: {rec}", pytrace=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def hiplib():
    """the product library; building it needs hipcc only (no GPU)"""
    from pagraph_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def run_group(cmd, timeout, **kw):
    """subprocess.run(cmd, capture_output=True, text=True, timeout=...) for commands that spawn GPU processes of their own
    (torch.distributed.run, the trainer scripts): the command gets a process GROUP, and on a time-out the whole group is
    killed before the test fails with what the command had printed. subprocess.run kills only its direct child: the ranks of
    a hung torchrun then stayed alive ON THE GPU, and minutes later the test process itself met a sticky
    hipErrorIllegalAddress and dumped core while pytest rendered that failure (round 4: both sightings of "the core dump"
    came right behind a timed-out two-rank bench; profiles/r04/core_dump_*.txt)."""
    import signal
    import subprocess
    import types
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        pytest.fail(f"timed out after {timeout} s (process group killed): {' '.join(map(str, cmd))[-300:]}\n"
                    f"--- stdout tail\n{out[-1500:]}\n--- stderr tail\n{err[-6000:]}", pytrace=False)
    finally:
        try:
            os.killpg(p.pid, signal.SIGKILL)        # stragglers of a command that returned (a rank that outlived its agent)
        except (ProcessLookupError, PermissionError):
            pass
    return types.SimpleNamespace(returncode=p.returncode, stdout=out, stderr=err)
