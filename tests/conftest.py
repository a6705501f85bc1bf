import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
GOLD = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gold():
    return GOLD


def pytest_collection_finish(session):
    """tests/test_gpu_dp.py needs two FRESH processes that share cuda:0 with this one. They are started here — after collection,
    before any test (hence before this process has made a single HIP call) — and joined by the test's fixture."""
    if session.config.option.collectonly or not any('test_gpu_dp' in item.nodeid for item in session.items):
        return
    if not Path('/dev/kfd').exists():          # no GPU on this machine; a device-file test makes no HIP call in this process
        return
    import socket
    import subprocess
    import tempfile
    tmp = Path(tempfile.mkdtemp(prefix='exorl_dp_'))
    data, out = tmp / 'buffer', tmp / 'out'
    out.mkdir()
    import _dp_worker
    _dp_worker.write_dataset(data)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        log = tmp / f'rank{rank}.log'
        procs.append((subprocess.Popen([sys.executable, str(ROOT / 'tests' / '_dp_worker.py'), str(data), str(out)], env=env,
                                       stdout=open(log, 'w'), stderr=subprocess.STDOUT), log))
    session.config._dp_children = {'procs': procs, 'out': out, 'data': data}


def _reap_dp_children(config):
    """The two rank processes outlive a run that never reaches test_gpu_dp's fixture (-x after an earlier failure, Ctrl-C): end them here."""
    st = getattr(config, '_dp_children', None)
    if not st:
        return
    import shutil
    for p, _ in st['procs']:
        if p.poll() is None:
            p.kill()
        try:
            p.wait(timeout=30)
        except Exception:
            pass
    shutil.rmtree(st['out'].parent, ignore_errors=True)
    config._dp_children = None


def pytest_sessionfinish(session, exitstatus):
    _reap_dp_children(session.config)


def pytest_unconfigure(config):
    _reap_dp_children(config)
