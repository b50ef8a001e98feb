#!/usr/bin/env python3
"""Build-time pin of the hand-off inside the merged trailing update (gemm.hip, gemm_f64_pipe_kernel<.., MERGED>).

The consumer on the look-ahead stream reads a producer tile as soon as the device counter says so, while the producer
kernel is still running.  That is only correct if, in the gfx950 ISA of the producer,
  (1) the 64 tile stores of a block-column workgroup are WRITE-THROUGH stores (global_store_dwordx2 ... sc1: the
      lowering of the agent-scope relaxed atomic stores in the source), not plain stores that may stay dirty in one
      XCD's L2,
  (2) they are followed by  s_waitcnt vmcnt(0)  and a barrier before
  (3) the single global_atomic_add that publishes the tile.
A compiler that lowered the atomic stores differently would corrupt the next panel silently, so the build
fails here instead.  Usage: check_isa.py build/gemm.s
"""
import re
import sys

MERGED_KERNELS = ['_Z20gemm_f64_pipe_kernelILb0ELb1ELb0ELb1EE',      # <BKM=false, LOWER=true, ZEROC=false, MERGED=true>: single GPU
                  '_Z20gemm_f64_pipe_kernelILb0ELb0ELb0ELb1EE']      # <.., LOWER=false, .., MERGED=true>: the driver's staircase


def main(path):
    text = open(path).read()
    for name in MERGED_KERNELS:
        check(text, name, path)


def check(text, MERGED, path):
    m = re.search(r'^(%s\w*):.*?^\s*\.amdhsa_kernel \1' % MERGED, text, re.S | re.M)
    if not m:
        sys.exit('check_isa: the merged trailing-update kernel (%s...) is not in %s' % (MERGED, path))
    body = m.group(0).split('\n')
    wt = [i for i, line in enumerate(body) if re.search(r'\bglobal_store_dwordx2\b.*\bsc1\b', line)]
    adds = [i for i, line in enumerate(body) if re.search(r'\bglobal_atomic_add\b', line)]
    if len(wt) != 64:
        sys.exit('check_isa: expected the 64 tile stores of the block-column branch as write-through (sc1) stores, found %d' % len(wt))
    if len(adds) != 1:
        sys.exit('check_isa: expected exactly one global_atomic_add (the tile counter), found %d' % len(adds))
    between = body[wt[-1]:adds[0]]
    if adds[0] < wt[-1] or not any(re.search(r's_waitcnt\s+vmcnt\(0\)', l) for l in between) or not any('s_barrier' in l for l in between):
        sys.exit('check_isa: the counter increment must follow the write-through stores, an s_waitcnt vmcnt(0) and a barrier')
    print('check_isa: %s: 64 sc1 stores -> s_waitcnt vmcnt(0) -> s_barrier -> global_atomic_add  OK' % MERGED)


if __name__ == '__main__':
    main(sys.argv[1])
