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
    assert bench.syrk_schedule_flops(32768) == 10320634183680.0
    assert bench.syrk_schedule_flops(32768, panels=1) == bench.syrk_launch_flops(32768)
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
