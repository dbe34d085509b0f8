import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def sw():
    import stringwars_amd
    return stringwars_amd


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle


@pytest.fixture(scope="session")
def scope(sw):
    """One GPU scope for the whole session; a missing device is a hard failure under `-m gpu`."""
    return sw.DeviceScope(gpu_device=0)


def child_pythonpath() -> str:
    """PYTHONPATH of a child process a test starts: the repo root IN FRONT of whatever this run inherited (never instead of it)."""
    inherited = os.environ.get("PYTHONPATH", "")
    return ROOT + (os.pathsep + inherited if inherited else "")


def run_in_child(request, env=None, test_library=False, timeout=1800) -> bool:
    """For tests that need an environment switch the library reads once per process, or a test hook that only the TEST build of the
    library carries (libstringwars_amd_test.so, -DSWH_TEST_HOOKS): the test runs AGAIN in a child process with that environment
    (and that library). Returns True in the child -- run the body -- and False in the parent, after the child has passed."""
    import subprocess
    if os.environ.get("SWH_TEST_CHILD"):     # a child never starts another one, whatever its node id looks like from where it runs
        return True
    # ROOT goes IN FRONT of the inherited PYTHONPATH (site hooks and paths the parent run relies on -- the driver's observation hook
    # among them -- stay in force in the child), and the child's test id is built from the test's file relative to ROOT, so it
    # collects the same test whatever directory the parent was started from
    child_env = dict(os.environ, SWH_TEST_CHILD=request.node.nodeid, PYTHONPATH=child_pythonpath(), **(env or {}))
    if test_library:
        child_env["STRINGWARS_AMD_LIBRARY"] = os.path.join(ROOT, "stringwars_amd", "libstringwars_amd_test.so")
    test_id = os.path.relpath(os.path.realpath(str(request.node.fspath)), os.path.realpath(ROOT)) + "::" + request.node.nodeid.split("::", 1)[1]
    done = subprocess.run([sys.executable, "-m", "pytest", test_id, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                          cwd=ROOT, env=child_env, capture_output=True, text=True, timeout=timeout)
    assert done.returncode == 0 and " passed" in done.stdout, done.stdout[-4000:] + done.stderr[-2000:]
    return False


TEST_LIBRARY_ENV = {"STRINGWARS_AMD_LIBRARY": os.path.join(ROOT, "stringwars_amd", "libstringwars_amd_test.so")}
