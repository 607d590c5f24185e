# -*- coding: utf-8 -*-
"""The drop-in boundary exercised from plain C: tests/cabi/cabi_smoke.c includes
include/ssq_hip.h, links libssq_hip.so and runs kernels on buffers it allocates through
the ABI's own runtime helpers -- no Python, no torch in the process."""
import os
import shutil
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(tmp_path, libdir, libfile):
    cc = shutil.which('gcc') or shutil.which('cc')
    assert cc, "no C compiler"
    assert os.path.isfile(os.path.join(libdir, libfile))
    exe = str(tmp_path / 'cabi_smoke')
    subprocess.check_call([cc, '-std=c99', '-O1', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'cabi', 'cabi_smoke.c'), '-o', exe,
                           '-L', libdir, '-l:' + libfile, '-lm',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'PASS' in out.stdout


def test_c_program_links_and_passes(tmp_path):
    if os.environ.get('SSQ_EMULATE') == '1':
        pytest.skip("see tests/test_cabi_symbols.py::test_c_client_against_emulated_library")
    build_and_run(tmp_path, os.path.join(ROOT, 'ssqueezepy_amd'), 'libssq_hip.so')


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device(tmp_path):
    """`bench.py --gpus 2` end to end on a one-GPU box: two ranks (torch.distributed.run started by
    bench.py itself, gloo, both on cuda:0 -- two RCCL ranks cannot share a device) shard the
    signals, run their own plans and close with the one all_gather of per-signal summaries. The
    N > 1 branch of the benchmark executes on hardware every round; RCCL itself needs N GPUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSQ_BENCH_BACKEND='gloo', SSQ_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
                          '--warmup', '1', '--batch', '4', '--no-cpu'],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['scaling'] == 'weak'
    assert line['per_gpu']['signals_per_step'] == 4 and line['value'] > 0


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_device_with_the_full_gather(tmp_path):
    """The driver's N = 8 line, rehearsed: `bench.py --gpus 8 --batch 1 --gather-tx` over gloo with all eight ranks on
    cuda:0 (round 6). What it checks is plumbing only -- rank arithmetic, each rank's share of the host threads, the
    summary table at N = 8 and `sharding.gather_tx` (one signal per rank, eight blocks received) -- so that the first
    run on eight GPUs cannot fail on any of it; no hardware claim follows from its numbers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSQ_BENCH_BACKEND='gloo', SSQ_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1',
                          '--warmup', '1', '--batch', '1', '--no-cpu', '--gather-tx', '--n', '20000', '--na', '100'],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['ranks'] == 8 and line['scaling'] == 'weak'
    assert line['per_gpu']['signals_per_step'] == 1 and line['value'] > 0
    g = line['with_full_tx_gather']
    assert g['gathered_bytes_per_rank'] == 7 * 1 * 100 * 20000 * 8 and g['transforms_per_s'] > 0
