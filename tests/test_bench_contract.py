"""bench.py's bookkeeping that can be checked without a GPU: flop formulas, workload tables, weak-scaling sizes."""
import math

import bench


def test_flop_formulas_match_survey_8d():
    # trailing-update flops at N=32768, nb=512 = 1.1455e13 (97.7 % of N^3/3)
    t = bench.trailing_flops(32768)
    assert abs(t - 1.1455e13) / 1.1455e13 < 1e-3
    assert abs(t / (32768**3 / 3.0) - 0.977) < 2e-3
    # the look-ahead split: syrk launches + next-panel column updates (t * nb * nb * 2 each, minus nothing) ~ the total
    assert bench.syrk_launch_flops(32768) < t and bench.syrk_launch_flops(32768) > 0.93 * t
    # the shipped schedule (two panels per update above 14336 trailing rows) moves a little more of it into the column
    # updates; profiles/r01_c3_bench.json records the library's own count of exactly this number
    assert bench.syrk_schedule_flops(32768, merged_min=None, tail_rows=0) == 10320634183680.0
    assert bench.syrk_schedule_flops(32768, panels=1, merged_min=None, tail_rows=0) == bench.syrk_launch_flops(32768)
    # the merged trailing update folds the next block column into the launch while >= 12288 rows trail: more of the total
    assert bench.syrk_schedule_flops(32768, merged_min=None, tail_rows=0) < bench.syrk_schedule_flops(32768) < t
    n, m = 32768, 4096
    assert bench.step_flops(n, m) == n**3 / 3.0 + 2.0 * n * n + float(n) * n * m + 2.0 * m * n


def test_workloads_and_weak_scaling_sizes():
    assert math.prod(bench.WORKLOADS['c2'][0]) == 8192 and math.prod(bench.WORKLOADS['c3'][0]) == 32768
    from mgpi_amd.distributed import choose_grid
    for g, (side, d, m, seed) in bench.MULTI.items():
        n = math.prod(side)
        P, Q = choose_grid(g)
        lcm = P * Q // math.gcd(P, Q)
        assert n % (bench.NB * lcm) == 0, (g, n)
        assert len(side) == d
        # weak scaling in the O(N^3) work: within 5 % of 32768 * g^(1/3)
        assert abs(n / (32768 * g ** (1.0 / 3.0)) - 1.0) < 0.05, (g, n)
    assert math.prod(bench.MULTI[8][0]) == 65536 and bench.MULTI[8][1] == 4      # BASELINE.json configs[3]


def test_inputs_are_seeded_and_shaped():
    x, f, xs, ell = bench.make_inputs((4, 4, 4), 10, 1)
    x2, f2, xs2, _ = bench.make_inputs((4, 4, 4), 10, 1)
    assert x.shape == (64, 3) and f.shape == (64,) and xs.shape == (10, 3) and len(ell) == 3
    assert (x == x2).all() and (f == f2).all() and (xs == xs2).all()
    assert ell == (0.5, 0.5, 0.5)


def test_gpus_flag_is_honoured_without_a_launcher():
    """`python bench.py --gpus N` with no launcher starts N ranks itself - or says why it cannot.  On this CPU-only
    box the device count (taken in a child process; the parent never touches the GPU) is 0, so the run must exit
    non-zero with a message instead of silently benchmarking one GPU."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), 'bench.py'), '--gpus', '2', '--steps', '1'],
                         env=env, capture_output=True, text=True, timeout=600)
    if bench.visible_gpus() >= 2:
        assert out.returncode == 0 and '"n_gpus": 2' in out.stdout, out.stderr[-2000:]
    else:
        assert out.returncode == 2 and '--gpus 2 requested' in out.stderr, (out.returncode, out.stderr[-2000:])
    # a launcher whose world size disagrees with --gpus is refused as well
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), 'bench.py'), '--gpus', '4', '--steps', '1'],
                         env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=600)
    assert out.returncode == 2 and 'launcher started 2' in out.stderr


def test_unique_id_rendezvous_between_processes(tmp_path):
    """rank 0's blob reaches every other rank through the rendezvous file (three real processes, no GPU)"""
    import os
    import subprocess
    import sys
    from mgpi_amd import rendezvous
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, sys\n'
        'sys.path.insert(0, %r)\n'
        'from mgpi_amd.rendezvous import exchange_unique_id, rank_env\n'
        'rank, local, world = rank_env()\n'
        'blob = exchange_unique_id(rank, world, lambda: bytes(range(128)), timeout=60)\n'
        'assert blob == bytes(range(128)) and world == 3 and local == rank\n'
        'open(os.path.join(%r, "ok%%d" %% rank), "w").close()\n'
        'import time\n'
        'while rank == 0 and not all(os.path.exists(os.path.join(%r, "ok%%d" %% r)) for r in range(world)): time.sleep(0.01)\n' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path), str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MGPI_RENDEZVOUS')}
    codes = rendezvous.spawn_ranks([str(script)], 3, env=env, timeout=120)
    assert codes == [0, 0, 0]
    assert all((tmp_path / ('ok%d' % r)).exists() for r in range(3))
    # the launcher-less naming (parent pid + master port): two children of THIS process agree on it
    script2 = tmp_path / 'rank2.py'
    for r in range(3):
        (tmp_path / ('ok%d' % r)).unlink()
    script2.write_text(script.read_text().replace('world == 3', 'world == 2'))
    procs = [subprocess.Popen([sys.executable, str(script2)], env=dict(env, RANK=str(r), WORLD_SIZE='2', MASTER_PORT='45678'))
             for r in (1, 0)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0]


def _rank_script(tmp_path, body):
    import os
    script = tmp_path / 'job.py'
    script.write_text('import os, sys, time\nsys.path.insert(0, %r)\n'
                      'from mgpi_amd.rendezvous import exchange_unique_id, rank_env\n'
                      'rank, local, world = rank_env()\n' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + body)
    return str(script)


def _clean_env():
    import os
    return {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MGPI_RENDEZVOUS')}


def test_spawn_ranks_reports_the_rank_that_died(tmp_path):
    """one rank exits 3 at once, the others would block for ever (a collective that lost its peer): the job ends within
    seconds, the survivors are killed, the exit codes and the dead rank's stderr come back (it used to hang)"""
    import io
    import time
    from mgpi_amd import rendezvous
    script = _rank_script(tmp_path, 'if rank == 2:\n    sys.stderr.write("rank two: device lost\\n"); sys.exit(3)\ntime.sleep(600)\n')
    report = io.StringIO()
    t0 = time.time()
    codes = rendezvous.spawn_ranks([script], 3, env=_clean_env(), timeout=300, report=report)
    assert time.time() - t0 < 60
    assert codes[2] == 3 and codes[0] != 0 and codes[1] != 0, codes
    text = report.getvalue()
    assert 'rank 2 exited with code 3' in text and 'rank two: device lost' in text


def test_spawn_ranks_deadline(tmp_path):
    """nobody fails, nobody finishes: the deadline ends the job with non-zero codes"""
    import io
    from mgpi_amd import rendezvous
    script = _rank_script(tmp_path, 'time.sleep(600)\n')
    report = io.StringIO()
    codes = rendezvous.spawn_ranks([script], 2, env=_clean_env(), timeout=3, report=report)
    assert all(c != 0 for c in codes) and 'no result within' in report.getvalue()


def test_unique_id_generations(tmp_path):
    """two exchanges in one job (a second communicator): the second returns the SECOND id on every rank, even when a slow
    rank 0 publishes it after the others already look for it (stale bytes made ncclCommInitRank hang)"""
    from mgpi_amd import rendezvous
    script = _rank_script(tmp_path,
                          'a = exchange_unique_id(rank, world, lambda: b"A" * 128, timeout=60)\n'
                          'if rank == 0: time.sleep(1.0)\n'
                          'b = exchange_unique_id(rank, world, lambda: b"B" * 128, timeout=60)\n'
                          'assert a == b"A" * 128 and b == b"B" * 128, (rank, a[:4], b[:4])\n'
                          'open(os.path.join(%r, "gen%%d" %% rank), "w").close()\n'
                          'while rank == 0 and not all(os.path.exists(os.path.join(%r, "gen%%d" %% r)) for r in range(world)): time.sleep(0.01)\n'
                          % (str(tmp_path), str(tmp_path)))
    codes = rendezvous.spawn_ranks([script], 3, env=_clean_env(), timeout=120)
    assert codes == [0, 0, 0]
