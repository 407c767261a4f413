"""The multi-GPU path's one collective on RCCL itself.  One GPU allows a world of one rank: `init_process_group("nccl")`,
`dist.gather_frames` both ways (all_gather / gather to rank 0), and one `bench.py --config 5` run inside the group
(barrier, max-over-ranks all_reduce and the timed result gather all go through RCCL).  Child processes, so a wedged
communicator cannot take the test session with it.  The N > 1 logic is covered by tests/test_dist_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GATHER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import mfcc_amd
from mfcc_amd import dist as md
from oracle import mfcc_float as mf
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
x = mf.synth_pcm(512 + 170 * 999, seed=3)
with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
    loc = m.process(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    for dst in (None, 0):
        full = md.gather_frames(loc, 13, dst=dst)
        torch.cuda.synchronize()
        assert full.is_cuda and tuple(full.shape) == (1000, 13) and torch.equal(full, loc), dst
    # the utterance plan on a group of one: everything is rank 0's
    idx, outs = md.process_corpus_sharded(m.process_batch, [x[:5000], x[5000:9000]], dist.get_rank(), dist.get_world_size())
    assert idx == [0, 1] and len(outs) == 2
ref = mf.mfcc_float_ref(x, n_cep=13)
assert np.abs(full.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
dist.barrier()
dist.destroy_process_group()
print("RCCL-WORLD1-OK")
"""


def _env():
    env = dict(os.environ)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_gather_frames_over_rccl_world_of_one():
    r = subprocess.run([sys.executable, "-c", GATHER % ROOT], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0 and "RCCL-WORLD1-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_bench_config5_step_inside_an_rccl_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", "1", "--group",
                        "--backend", "nccl", "--utterances", "1250", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    b = json.loads(r.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 1 and b["scaling"] == "strong" and b["config"]["frames_per_step"] == 1250 * 939
    assert b["gather"] and b["gather"]["backend"] == "nccl" and b["gather"]["bytes"] == 1250 * 939 * 13 * 4
    he = b["host_enqueue"]["this_rank"]
    assert he["utterances"] == 1250 and he["enqueue_us_per_step"] > 0
