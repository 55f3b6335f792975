"""GPU: bench.py's multi-rank path rehearsed on one GPU.  The driver launches `bench.py --gpus N` over
RCCL on an 8-GPU node; this box has one GPU, so the N ranks share it and exchange their tiles over gloo (SRT_BENCH_REHEARSE=1):
the same tile sharding, the same gather layout, the same un-tiling - and the record must verify itself: the epoch image of the
rehearsal equals the image the REFERENCE build rendered (tests/golden/pt_fullsize.json, cfg4_cbox_1024_64spp)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import _harness as H

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks", [2])
def test_bench_two_ranks_rehearsal_reproduces_the_reference_image(ranks):
    env = dict(os.environ, SRT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--no-cfg5", "--no-raster", "--no-elision", "--no-cpu-baseline", "--no-dropin"]
    # a FRESH child process (never an exec from a process that has touched the GPU)
    p = subprocess.run(cmd, env=env, cwd=H.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["image_equals_golden"] is True, d["golden_check"]
    assert len(d["rank_kernel_ms"]) == ranks and all(ms > 0 for ms in d["rank_kernel_ms"])
    assert sum(d["rank_rays"]) == d["rays"] and d["gather_ms"] is not None
    assert d["collective"]["ranks"] == ranks and d["collective"]["is_rccl"] is False          # the rehearsal says what it is
