"""The data-parallel exchange on the GPU box: a fresh 1-rank torch.distributed.run child over RCCL (tools/dp_selfcheck.py).
One GPU per box here, so this covers the engine hook -> early decoder all-reduce -> CU reservation -> finish path, not scaling."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize('model', ['nano224', 'gpt2_lora', 'gpt2_lora:deterministic', 'nano224:deterministic', 'gpt2_lora:gloo', 'gpt2_lora:gloo+torch'])
def test_one_rank_rccl_hooked_exchange_matches_unhooked(model):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    model, _, mode = model.partition(':')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', I2T_DP_SELFCHECK_MODEL=model)
    if mode == 'deterministic':     # fixed-order reductions: hooked == un-hooked and accumulated == accumulated EXACTLY (tolerance 0)
        env['I2T_DETERMINISTIC'] = '1'
    if mode.startswith('gloo'):          # torch.distributed on gloo as the control plane (bench.py / train_loop): the package's communicator is the
        env['I2T_DP_SELFCHECK_PG'] = 'gloo'      # process's only RCCL communicator; '+torch': the fallback exchange on a torch NCCL group
        if mode.endswith('+torch'):
            env['I2T_DP_COMM'] = 'torch'
    env.pop('NCCL_MAX_NCHANNELS', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tools', 'dp_selfcheck.py')]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=540)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0 and 'DP_SELFCHECK_OK' in out, out[-4000:]
    assert 'nchannels=16' in out or os.environ.get('I2T_RCCL_CUS'), out[-500:]
    assert ('transport=torch-nccl' if mode.endswith('+torch') else 'transport=rccl-abi') in out, out[-800:]
    assert f'deterministic={int(mode == "deterministic")}' in out, out[-800:]        # the exchange ran on the C-ABI communicator (include/i2t.h i2t_comm_*)
