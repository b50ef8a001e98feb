"""The pure index arithmetic of the tile kernels (csrc/tile_index.h: tile orders, the block-cyclic staircase, the
update-role enumeration of the panel step, the buffer pool's choice) built for the CPU with AddressSanitizer and
UndefinedBehaviorSanitizer and run exhaustively over small shapes - SURVEY.md section 5 "sanitizers".  GPU ASan is not
available on the pool; this covers the host-reachable part of the code that decides which memory a workgroup touches."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, 'muses-gaussian-process-interpolator_amd', 'csrc')


@pytest.mark.skipif(shutil.which('g++') is None, reason='no g++')
def test_index_functions_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'index_test')
    build = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                            '-I', CSRC, os.path.join(CSRC, 'index_test.cpp'), '-o', exe], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1', UBSAN_OPTIONS='print_stacktrace=1'))
    assert run.returncode == 0 and 'INDEX-OK' in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
