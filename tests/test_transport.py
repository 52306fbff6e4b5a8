"""Record transport between slab neighbours over gloo: the agreed-size messages with a count header,
the second message when a count outgrows the agreed size, empty messages, both directions differing."""
import os
import subprocess
import sys

import pytest

from test_slabs_cpu import free_port

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_with_jumping_counts(world):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "transport_worker.py"), "--rank", str(r),
                               "--world", str(world), "--port", str(port)]) for r in range(world)]
    try:
        codes = [p.wait(timeout=120) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world


@pytest.mark.gpu
def test_exchange_over_rccl_with_device_buffers():
    """the "nccl" branch of the transport (device buffers handed to RCCL directly), one rank talking to itself"""
    port = free_port()
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "transport_worker.py"), "--rank", "0", "--world", "1",
                          "--port", str(port), "--rccl-self"])
    try:
        code = p.wait(timeout=100)
    finally:
        if p.poll() is None:
            p.kill()
    assert code == 0
