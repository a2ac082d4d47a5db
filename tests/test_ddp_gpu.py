"""SURVEY.md 8e end to end on the GPU: two ranks (gloo rendezvous, both on the box's single GPU) through
DistributedDataParallelHIP, the globally normalised criterion and the fused optimizer -- see
tests/helpers/ddp_parity_worker.py for what is asserted."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_two_ranks_match_single_process_replay():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "ddp_parity_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("DDP_PARITY")]
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-3000:])
    print(line[0])
