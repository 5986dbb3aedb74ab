"""bench.py's reporting contract, on CPU: the ONE stdout line stays below the driver's tail (round 2 lost its headline
because the line had grown to 20 KB), never carries a roofline fraction above 1, and `python bench.py --gpus N`
launches its own ranks (train.py:307 mp.spawn in the reference) or refuses cleanly."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def _full_line():
    """The full round-2 report (everything bench.py gathered, 20 KB) as the worst-case input."""
    with open(os.path.join(ROOT, 'profiles', 'r02_bench_line.json')) as f:
        text = f.read()
    return json.loads(text[text.index('{'):])


def test_compact_line_fits_the_driver_tail():
    import bench
    full = _full_line()
    assert len(json.dumps(full)) > 15000
    full['config']['lanes'] = 2
    line = bench.compact_line(full, 'gpurun_out/bench_detail.json')
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT_BYTES
    for key in bench.CONTRACT_KEYS:
        assert key in line, key
    assert line['roofline']['frac'] <= 1.0
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(line['roofline'])
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline'])
    assert line['parity']['ok'] is True and line['detail'] == 'gpurun_out/bench_detail.json'
    assert 'hip_kernels' not in line and 'roofline_rows' not in line and 'census' not in line


def test_compact_line_survives_long_free_text():
    import bench
    full = _full_line()
    full['cpu_baseline']['sample'] = 'x' * 5000
    full['config']['lanes_note'] = 'y' * 3000
    assert len(json.dumps(bench.compact_line(full, None))) < bench.LINE_LIMIT_BYTES


def test_compact_line_rejects_fraction_above_one():
    import bench
    full = _full_line()
    full['roofline']['frac'] = 1.65
    with pytest.raises(AssertionError):
        bench.compact_line(full, None)


def test_self_launch_refuses_oversubscription(capfd):
    import bench
    assert bench.launch_ranks(2, [], device_count=1) == 2
    assert 'refusing to oversubscribe' in capfd.readouterr().err


@pytest.mark.timeout(300)
def test_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the process launches two ranks, they rendezvous on 127.0.0.1,
    and exactly one line comes out of rank 0 (the rendezvous-only mode needs no GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check'],
                         capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {'launch_check': True, 'n_gpus': 2, 'rank_sum': 3.0}


@pytest.mark.timeout(300)
def test_cli_refuses_more_ranks_than_gpus():
    """The command-line path of the refusal: exit code 2, nothing started."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'], capture_output=True, text=True,
                         env=env, timeout=280)
    assert out.returncode == 2 and 'refusing to oversubscribe' in out.stderr


def test_time_budget_helper():
    """The legs after the timed region run under a deadline: a leg that overruns is abandoned with a reason (the line is
    printed without it), one that finishes hands its result over, one that raises raises on the caller's thread."""
    import time
    import bench
    assert bench.run_with_deadline(lambda: 7, 5.0) == (7, None)
    got, why = bench.run_with_deadline(lambda: time.sleep(30), 1.0)
    assert got is None and 'time budget' in why
    with pytest.raises(ZeroDivisionError):
        bench.run_with_deadline(lambda: 1 / 0, 5.0)


def test_compact_line_round4_fields():
    """Round 4: the line of the round (profiles/r04_a_bench_line.json) names a VALU-bound kernel when it dominates, carries
    `roofline.worst`, `side_configs` and `cpu_baseline.os_cpu_count`, and the compaction keeps all of them below the limit."""
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r04_a_bench_line.json')) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    assert line['roofline']['bound'] in ('hbm', 'mfma', 'valu', 'latency') and 0 < line['roofline']['frac'] <= 1.0
    assert set(line['roofline']['worst']) == {'kernel', 'bound', 'frac', 'ms_per_step'}
    assert set(line['side_configs']) == {'camlipwc', 'kitti'} and line['cpu_baseline']['os_cpu_count'] > 0
    full = dict(_full_line(), side_configs=line['side_configs'])
    full['roofline']['worst'] = line['roofline']['worst']
    full['cpu_baseline']['os_cpu_count'] = 256
    out = bench.compact_line(full, 'gpurun_out/bench_detail.json')
    assert out['side_configs'] == line['side_configs'] and out['roofline']['worst'] == line['roofline']['worst']
    assert out['cpu_baseline']['os_cpu_count'] == 256 and len(json.dumps(out)) < bench.LINE_LIMIT_BYTES


def test_roofline_report_lets_every_kind_compete():
    """camli_knn (valu) could never be the roofline kernel.  Now the entry point with the most device time wins
    whatever bounds it, and `worst` is the lowest fraction among those holding >= 1 % of the step."""
    import types
    import bench
    summary = {
        'camli_knn': {'total_ms': 40.0, 'launches': 170, 'work': 170 * 5.0e7, 'unit': 'pairs', 'flop': 0.0},
        'camli_pointconv_dw_fwd': {'total_ms': 30.0, 'launches': 540, 'work': 540 * 1.2e8, 'unit': 'B', 'flop': 0.0},
        'camli_fps': {'total_ms': 25.0, 'launches': 5, 'work': 5 * 16 * 8192 * 4096.0, 'unit': 'point-updates', 'flop': 5 * 4096.0},
        'camli_allpairs_build_fwd': {'total_ms': 20.0, 'launches': 5, 'work': 5 * 3.0e9, 'unit': 'B', 'flop': 5 * 3.6e11},
    }
    args = types.SimpleNamespace(batch=8, iters=12, height=540, width=960, points=8192)
    roof, table = bench.roofline_report(summary, 5, args, step_ms=233.0)
    assert roof['kernel'] == 'camli_knn' and roof['bound'] == 'valu' and roof['unit'] == 'Gpairs/s'
    assert abs(roof['frac'] - (170 * 5.0e7 / 0.040 / 1e9) / bench.VALU_PAIR_PEAK_G) < 1e-3
    assert table['camli_fps']['frac'] == round(bench.FPS_STEP_IDEAL_US / (25.0e3 / (5 * 4096.0)), 4)
    assert roof['worst']['kernel'] == min(table, key=lambda n: table[n]['frac'])
    # with the KNN entry gone the FPS launch dominates: a latency-bound roofline object
    del summary['camli_knn']
    summary['camli_fps']['total_ms'] = 50.0
    roof, _ = bench.roofline_report(summary, 5, args, step_ms=233.0)
    assert roof['kernel'] == 'camli_fps' and roof['bound'] == 'latency' and roof['frac'] <= 1.0


def test_compact_line_round5_fields():
    """Round 5: the line says where the STEP stands (`roofline.step`: matrix-core flop, streamed bytes, floor,
    fraction), carries the single-lane fraction as a SCALAR next to the object (a parser that keeps scalars dropped the object),
    names where `traffic` comes from, prices configs[3]'s per-rank step (`side_configs.ddp4`) and keeps the all-physical-cores
    CPU point -- all below the limit."""
    import bench
    full = _full_line()
    full['roofline'].update(traffic_source='stored PMC pass r03 (profiles/roofline_traffic.json), not this run', frac_single_lane=0.5,
                            step={'mfma_flop': 21800000000000, 'library_flop': 17000000000000, 'hbm_bytes': 13300000000,
                                  'floor_ms': 140.3, 'frac': 0.6452})
    full['side_configs'] = {'camlipwc': {'ms_per_step': 62.4, 'value': 16.0, 'dtype': 'f32', 'steps': 5},
                            'kitti': {'ms_per_step': 124.9, 'value': 8.0, 'dtype': 'bf16', 'steps': 5},
                            'ddp4': {'ms_per_step': 120.0, 'value': 33.3, 'dtype': 'f32', 'steps': 5, 'batch': 4, 'n_iters': 12,
                                     'sync_bn': True, 'ranks': 1, 'collectives': '1-rank RCCL group'}}
    full['cpu_baseline']['all_cores'] = {'value': 0.02, 'cores': 64, 'seconds_per_step': 50.0, 'sample': 'one batch-1 training step, no warm-up'}
    line = bench.compact_line(full, 'gpurun_out/bench_detail.json')
    assert len(json.dumps(line)) < bench.LINE_LIMIT_BYTES
    roof = line['roofline']
    assert set(roof['step']) == {'mfma_flop', 'library_flop', 'hbm_bytes', 'floor_ms', 'frac'} and 0 < roof['step']['frac'] <= 1
    assert isinstance(roof['frac_single_lane'], float) and 'stored PMC pass' in roof['traffic_source']
    assert set(line['side_configs']) == {'camlipwc', 'kitti', 'ddp4'} and line['side_configs']['ddp4']['batch'] == 4
    assert line['cpu_baseline']['all_cores']['cores'] == 64
    assert 'ddp4' in bench.SIDE_CONFIGS


def test_roofline_round6_fields():
    """Round 6: `frac` is the single-lane (rocprofv3-comparable) figure and `frac_in_situ` the two-lane one;
    `roofline.north_star` names the dominant SURVEY 8(a) kernel when the overall dominant one is a convolution-side (8(f)2)
    entry point; the Winograd entry points are priced as matrix-core kernels; ddp4 carries its host enqueue time."""
    import types
    import bench
    summary = {
        'camli_wino_conv3x3': {'total_ms': 60.0, 'launches': 200, 'work': 200 * 7.0e8, 'unit': 'B', 'flop': 200 * 2.5e10},
        'camli_convcl_fwd': {'total_ms': 20.0, 'launches': 48, 'work': 48 * 1.5e8, 'unit': 'B', 'flop': 48 * 3.2e10},
        'camli_pointconv_dw_fwd': {'total_ms': 30.0, 'launches': 540, 'work': 540 * 1.2e8, 'unit': 'B', 'flop': 0.0},
        'camli_knn': {'total_ms': 10.0, 'launches': 170, 'work': 170 * 5.0e7, 'unit': 'pairs', 'flop': 0.0},
    }
    args = types.SimpleNamespace(batch=8, iters=12, height=540, width=960, points=8192)
    roof, table = bench.roofline_report(summary, 5, args, step_ms=200.0)
    assert roof['kernel'] == 'camli_wino_conv3x3' and roof['bound'] == 'mfma' and 0 < roof['frac'] <= 1
    star = roof['north_star']
    assert star['kernel'] == 'camli_pointconv_dw_fwd' and star['bound'] == 'hbm' and star['frac_in_situ'] == table['camli_pointconv_dw_fwd']['frac']
    assert bench.CONV_SIDE <= set(bench.NORTH_STAR) and 'camli_knn' not in bench.CONV_SIDE
    assert abs(bench.kernel_frac('camli_knn', summary['camli_knn']) - table['camli_knn']['frac']) < 1e-3
    full = _full_line()
    full['roofline'].update(frac=0.74, frac_in_situ=0.45, avg_launch_us=275.0, avg_launch_us_in_situ=448.0, north_star=dict(star, frac=0.4, avg_launch_us=50.0))
    full['side_configs'] = {'ddp4': {'ms_per_step': 120.0, 'value': 33.3, 'dtype': 'f32', 'steps': 5, 'batch': 4, 'host_enqueue_ms': 118.0}}
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < bench.LINE_LIMIT_BYTES
    assert line['roofline']['frac_in_situ'] == 0.45 and line['roofline']['north_star']['kernel'] == 'camli_pointconv_dw_fwd'
    assert line['side_configs']['ddp4']['host_enqueue_ms'] == 118.0
    assert 'camli_corr3d_cost_levels_fwd' not in bench.NORTH_STAR          # entry points removed in round 5


def test_pmc_traffic_names_its_source():
    import types
    import bench
    args = types.SimpleNamespace(batch=8, iters=12, height=540, width=960, points=8192)
    traffic, source = bench.pmc_traffic('camli_pointconv_dw_fwd', args)
    assert traffic and 'stored PMC pass' in source and 'not this run' in source
    assert bench.pmc_traffic('no_such_entry_point', args) == (None, None)
